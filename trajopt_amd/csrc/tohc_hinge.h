// tohc_hinge.h -- SURVEY.md 8 (f2): the collision hinges of one linearization, assembled on the device in the column
// layout BPMPD takes, so that the SQP host reads constraint rows instead of contacts.
//
// What the reference does on the host with every contact, per SQP iteration (all fp64):
//   CollisionsToDistanceExpressions   src/trajopt/collision_terms.cpp:37-80    dist(x) ~ const + grad . x  (cast: blended by time)
//   CollisionCost::convex             collision_terms.cpp:205-214              viol = dist_pen - dist(x);  addHinge(viol, coeff)
//   ConvexObjective::addHinge         src/sco/modeling.cpp:22-29               new var h >= 0, row  viol - h <= 0, objective += coeff * h
//   BPMPDModel::optimize              src/sco/bpmpd_interface.cpp:320-372      rhs = -constant; per VARIABLE the (row, value) list,
//                                                                              simplify2'd (solver_interface.cpp:23-37: exact zeros
//                                                                              dropped, rows ascending) = acolcnt / acolidx / acolnzs
// Here: the slab is put in canonical order (problem, t, discrete before cast, geometry ids), row i of a problem = its i-th
// contact, and for every trajectory variable x[t*K + k] of every problem the kernel writes the column's (row, value) list:
//   discrete contact at t        value = -grad0[k]
//   cast contact (t, t+gap)      value = -((1 - time) * grad0[k])  in column (t, k),   -(time * grad1[k])  in column (t+gap, k)
// with exactly the reference's operation order (`1 - time` is float arithmetic, collision_terms.cpp:73), so the numbers are
// bit-identical to what sco would hand BPMPD.  The hinge variables' own columns (one entry -1 each), their bounds
// [0, BIG] and objective entries `coeff` are constants the host adds (include/trajopt_hipcc.h: tohc_hinges).
#pragma once
#include <stdint.h>

#include "../../include/trajopt_hipcc.h"

namespace tohc {

struct HingeArgs {
  const tohc_record *rec;   // the slab as k_finish wrote it (unordered) ...
  const double *grads;      // ... [n][2][K]
  const uint32_t *idx;      // canonical position -> slab position (radix sort of the 64-bit keys)
  const uint64_t *keys;     // the sorted keys: [set 21][cast 1][child A 21][child B 21]
  unsigned int n;
  int B, T, K, gap;
  double pen_disc, pen_cast;
  double *rhs, *dist, *wa, *wb;   // per row: -constant, distance, blend weights of the two halves
  int32_t *row_t;
  unsigned int *seg;        // [B*T*2 + 1] first row of every (problem, t, discrete | cast) segment
  long long *row_off;       // [B + 1]
  unsigned int *colcnt;     // [B*T*K]
  long long *col_off;       // [B*T*K + 1]
  int32_t *col_idx;
  double *col_val;
};

__global__ void __launch_bounds__(256) k_hinge_rows(HingeArgs a) {
  const unsigned int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const tohc_record &r = a.rec[a.idx[i]];
  const bool cast = r.cctype != TOHC_CC_NONE;
  double wa = 1.0, wb = 0.0, c = r.const0;
  if (cast) {   // collision_terms.cpp:71-77: exprScale(exprs0, 1 - time), exprScale(exprs1, time), sum
    const float tf = r.time;
    wa = (double)(1.0f - tf);
    wb = (double)tf;
    c = wa * r.const0 + wb * r.const1;
  }
  const double viol = (cast ? a.pen_cast : a.pen_disc) - c;   // exprSub(AffExpr(dist_pen), expr), collision_terms.cpp:210
  a.rhs[i] = -viol;                                            // rhs[iCnt] = -aff.constant, bpmpd_interface.cpp:361
  a.dist[i] = r.dist;
  a.wa[i] = wa; a.wb[i] = wb;
  a.row_t[i] = r.t | (cast ? TOHC_HINGE_CAST : 0);
}

// first row of every (problem, t, kind) segment: lower bound of the segment number in the sorted keys
__global__ void __launch_bounds__(256) k_hinge_seg(HingeArgs a) {
  const unsigned int s = blockIdx.x * 256 + threadIdx.x;
  const unsigned int nseg = (unsigned int)(a.B * a.T * 2);
  if (s > nseg) return;
  unsigned int lo = 0, hi = a.n;
  while (lo < hi) {
    const unsigned int mid = (lo + hi) >> 1;
    if ((a.keys[mid] >> 42) < (uint64_t)s) lo = mid + 1; else hi = mid;
  }
  a.seg[s] = lo;
  if (s % (unsigned int)(2 * a.T) == 0) a.row_off[s / (unsigned int)(2 * a.T)] = (long long)lo;
}

// The columns of one (problem, t): K trajectory variables that share their rows -- the cast contacts of (t - gap, t)
// [second half of the gradient row], the discrete contacts at t, the cast contacts of (t, t + gap) [first half], in that
// (= ascending row) order.  ONE BLOCK per (problem, t): thread (r, k) takes row i0 + r and column k, so the K doubles of a
// gradient row are read by K consecutive threads (coalesced); the position of an entry in its column is the number of
// non-zeros in the column's earlier rows: a running count per column in LDS + the flags of the rows above it in this
// round (both double-buffered: one barrier per round).  FILL = false counts, FILL = true writes.
#define TOHC_HINGE_MAXK 64
template <bool FILL>
__global__ void __launch_bounds__(256) k_hinge_cols(HingeArgs a) {
  __shared__ unsigned int base[2][TOHC_HINGE_MAXK];
  __shared__ unsigned char flag[2][256];
  const int K = a.K, RP = 256 / K;   // rows per round
  const unsigned int pt = blockIdx.x;
  const int t = (int)(pt % (unsigned int)a.T), p = (int)(pt / (unsigned int)a.T);
  const int tid = (int)threadIdx.x, r = tid / K, k = tid - r * K;
  const bool lane_on = r < RP;
  if (tid < K) base[0][tid] = 0;
  const long long row0 = a.row_off[p];
  const size_t c = (size_t)pt * K + (lane_on ? k : 0);
  const long long out0 = (FILL && lane_on) ? a.col_off[c] : 0;
  const size_t W = (size_t)2 * K;
  int round = 0;
  for (int part = 0; part < 3; ++part) {
    const bool second = part == 0;
    if (part == 0 && !(a.gap > 0 && t - a.gap >= 0)) continue;
    const unsigned int s = part == 0 ? (unsigned int)((p * a.T + t - a.gap) * 2 + 1) : (unsigned int)((p * a.T + t) * 2 + (part - 1));
    const unsigned int b = a.seg[s], e = a.seg[s + 1];
    for (unsigned int i0 = b; i0 < e; i0 += (unsigned int)RP, ++round) {
      const unsigned int i = i0 + (unsigned int)r;
      const bool valid = lane_on && i < e;
      double v = 0;
      if (valid) {
        const double w = second ? a.wb[i] : a.wa[i];
        const double g = a.grads[(size_t)a.idx[i] * W + (second ? K : 0) + k];
        v = -(w * g);
      }
      const bool nz = v != 0;   // simplify2: `if (vals[i] != 0)`, solver_interface.cpp:27
      const int cur = round & 1;
      flag[cur][tid] = nz ? 1 : 0;
      __syncthreads();
      if (lane_on) {
        unsigned int above = 0;
        for (int q = 0; q < r; ++q) above += flag[cur][q * K + k];
        const unsigned int pos = base[cur][k] + above;
        if (FILL && nz) { a.col_idx[out0 + pos] = (int32_t)((long long)i - row0); a.col_val[out0 + pos] = v; }
        if (r == RP - 1) base[cur ^ 1][k] = pos + (nz ? 1u : 0u);   // the last row lane has seen the whole round
      }
    }
  }
  __syncthreads();
  if (!FILL && tid < K) a.colcnt[(size_t)pt * K + tid] = base[round & 1][tid];
}

// K > TOHC_HINGE_MAXK: one thread per trajectory variable walks its column's rows
template <bool FILL>
__global__ void __launch_bounds__(256) k_hinge_cols_wide(HingeArgs a) {
  const unsigned int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= (unsigned int)(a.B * a.T * a.K)) return;
  const int k = (int)(c % (unsigned int)a.K);
  const unsigned int pt = c / (unsigned int)a.K;
  const int t = (int)(pt % (unsigned int)a.T), p = (int)(pt / (unsigned int)a.T);
  const long long row0 = a.row_off[p];
  long long out = FILL ? a.col_off[c] : 0;
  unsigned int cnt = 0;
  const size_t W = (size_t)2 * a.K;
  auto range = [&](unsigned int s, bool second) {
    for (unsigned int i = a.seg[s], e = a.seg[s + 1]; i < e; ++i) {
      const double w = second ? a.wb[i] : a.wa[i];
      const double g = a.grads[(size_t)a.idx[i] * W + (second ? a.K : 0) + k];
      const double v = -(w * g);
      if (v != 0) {
        if (FILL) { a.col_idx[out + cnt] = (int32_t)((long long)i - row0); a.col_val[out + cnt] = v; }
        ++cnt;
      }
    }
  };
  if (a.gap > 0 && t - a.gap >= 0) range((unsigned int)((p * a.T + t - a.gap) * 2 + 1), true);
  range((unsigned int)((p * a.T + t) * 2), false);
  range((unsigned int)((p * a.T + t) * 2 + 1), false);
  if (!FILL) a.colcnt[c] = cnt;
}

// exclusive prefix sum of the column counts (one block, tiles of 1024 consecutive counts: coalesced; ~1e5 columns)
__global__ void __launch_bounds__(1024) k_hinge_scan(const unsigned int *__restrict__ cnt, unsigned int n, long long *__restrict__ off) {
  __shared__ long long wsum[16];
  __shared__ long long carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long carry = 0;
  for (unsigned int t0 = 0; t0 < n; t0 += 1024) {
    const unsigned int i = t0 + threadIdx.x;
    const long long x = i < n ? (long long)cnt[i] : 0;
    long long v = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long o = __shfl_up(v, d);
      if (lane >= d) v += o;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    long long before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < n) off[i] = carry + before + v - x;
    if (threadIdx.x == 1023) carry_s = carry + before + v;
    __syncthreads();
    carry = carry_s;
  }
  if (threadIdx.x == 0) off[n] = carry;
}

}  // namespace tohc
