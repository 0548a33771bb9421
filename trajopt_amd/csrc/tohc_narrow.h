// tohc_narrow.h -- narrowphase device code: GJK closest points, GJK2+EPA penetration, swept-hull support, cast contact
// classification.  One GROUP of G lanes (G = 16 or 64, a power of two <= the 64-lane wavefront) works on one
// (timestep, query child, other child) pair:
//   * hull vertices live in REGISTERS, vertex i in lane i%G slot i/G; a support query is a lane-parallel dot product
//     followed by a cross-lane arg-max with lowest-index tie-break (btVector3::maxDot is a strict '>' scan from 0);
//   * the simplex / EPA bookkeeping is group-uniform: every lane of the group runs the same scalar control flow on
//     the same values, so no lane ever reads what another lane wrote -- except through the arg-max / broadcast;
//   * EPA's face and vertex pools live in LDS (EpaMem, one per group).
// Arithmetic follows Bullet's fp32 operation order line by line (the reference's vendored, patched Bullet):
//   [gjk] ext/bullet/src/BulletCollision/NarrowPhaseCollision/btGjkPairDetector.cpp:82-452
//   [vss] .../btVoronoiSimplexSolver.cpp:35-608            [epa] .../btGjkEpa2.cpp:142-945
//   [pds] .../btGjkEpaPenetrationDepthSolver.cpp:24-64     [cvx] .../CollisionShapes/btConvexShape.cpp:64-308
//   [bcc] src/trajopt/bullet_collision_checker.cpp:98-131 (GetAverageSupport), :816-882 (CastHullShape), :894-972
// Margins: boxes, hulls and cylinders carry margin 0 (MARGIN, bcc:23), so for them the "+ margin * n" terms of the
// reference are exact no-ops and are skipped; a sphere is the point at its origin with margin = radius
// (btSphereShape.h:62-67), and every margin term of the reference is applied when a margin is not zero.
#pragma once
#include <float.h>
#include <limits.h>
#include <stdint.h>

#include "tohc_math.h"

namespace tohc {

// ---------------------------------------------------------------------------------------------------------------
// group primitives
// ---------------------------------------------------------------------------------------------------------------
template <int G>
struct Grp {
#if defined(__HIPCC__)
  static TOHC_DEV int lane() { return (int)(threadIdx.x & (G - 1)); }
  static TOHC_DEV float bcast(float v, int src) { return __shfl(v, src, G); }
  // the same with the source lane known at compile time: inside a quad that is ONE DPP move, no trip through the LDS crossbar
  template <int SRC>
  static TOHC_DEV float bcastC(float v) {
    if constexpr (G == 4) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), SRC * 0x55, 0xf, 0xf, false));
    else return __shfl(v, SRC, G);
  }
  // lexicographic (max value, min index) all-reduce.  Inside a 16-lane DPP row the butterfly is four single VALU
  // DPP moves (quad_perm xor-1, quad_perm xor-2, row_half_mirror, row_mirror): no LDS crossbar round trips.
  static TOHC_DEV void combine(float &v, int &i, float ov, int oi) {
    bool take = (ov > v) || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
  }
  template <int CTRL>
  static TOHC_DEV void dppStep(float &v, int &i) {
    int vi = __float_as_int(v);
    float ov = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, CTRL, 0xf, 0xf, false));
    int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false);
    combine(v, i, ov, oi);
  }
  static TOHC_DEV void argmax(float &v, int &i) {
    if (G >= 2) dppStep<0xB1>(v, i);    // quad_perm [1,0,3,2]
    if (G >= 4) dppStep<0x4E>(v, i);    // quad_perm [2,3,0,1]
    if (G >= 8) dppStep<0x141>(v, i);   // row_half_mirror
    if (G >= 16) dppStep<0x140>(v, i);  // row_mirror
#pragma unroll
    for (int m = 16; m < G; m <<= 1) combine(v, i, __shfl_xor(v, m, G), __shfl_xor(i, m, G));
  }
  // (max value, MAX index) all-reduce
  static TOHC_DEV void combineHi(float &v, int &i, float ov, int oi) {
    bool take = (ov > v) || (ov == v && oi > i);
    v = take ? ov : v;
    i = take ? oi : i;
  }
  template <int CTRL>
  static TOHC_DEV void dppStepHi(float &v, int &i) {
    int vi = __float_as_int(v);
    float ov = __int_as_float(__builtin_amdgcn_update_dpp(vi, vi, CTRL, 0xf, 0xf, false));
    int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false);
    combineHi(v, i, ov, oi);
  }
  // bitwise OR over the group's lanes, the result in every lane (the DPP butterfly of argmax)
  template <int CTRL>
  static TOHC_DEV unsigned dppOr(unsigned v) { return v | (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
  static TOHC_DEV unsigned orAll(unsigned v) {
    if (G >= 2) v = dppOr<0xB1>(v);
    if (G >= 4) v = dppOr<0x4E>(v);
    if (G >= 8) v = dppOr<0x141>(v);
    if (G >= 16) v = dppOr<0x140>(v);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v |= (unsigned)__shfl_xor((int)v, m, G);
    return v;
  }
  static TOHC_DEV unsigned long long orAll64(unsigned long long v) {
    return ((unsigned long long)orAll((unsigned)(v >> 32)) << 32) | orAll((unsigned)v);
  }
  // does any lane of the group say so?  (every lane of the group must be active)
  static TOHC_DEV bool any(bool p) {
    const unsigned long long b = __ballot(p);
    if (G >= 64) return b != 0;
    const unsigned base = (unsigned)(threadIdx.x & 63) & ~(unsigned)(G - 1);
    return ((b >> base) & ((1ull << (G & 63)) - 1ull)) != 0;
  }
  static TOHC_DEV void fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }   // LDS written by one lane, read by another of the wavefront
  static TOHC_DEV void argmax_hi(float &v, int &i) {
    if (G >= 2) dppStepHi<0xB1>(v, i);
    if (G >= 4) dppStepHi<0x4E>(v, i);
    if (G >= 8) dppStepHi<0x141>(v, i);
    if (G >= 16) dppStepHi<0x140>(v, i);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) combineHi(v, i, __shfl_xor(v, m, G), __shfl_xor(i, m, G));
  }
#else
  static int lane() { return 0; }
  static float bcast(float v, int) { return v; }
  static void argmax(float &, int &) {}
  static void argmax_hi(float &, int &) {}
  static unsigned long long orAll64(unsigned long long v) { return v; }
  static bool any(bool p) { return p; }
  static void fence() {}
#endif
};

#if defined(__HIPCC__)
template <>
struct Grp<1> {  // one lane per pair: nothing to exchange
  static TOHC_DEV int lane() { return 0; }
  static TOHC_DEV float bcast(float v, int) { return v; }
  static TOHC_DEV void argmax(float &, int &) {}
  static TOHC_DEV void argmax_hi(float &, int &) {}
  static TOHC_DEV unsigned long long orAll64(unsigned long long v) { return v; }
  static TOHC_DEV bool any(bool p) { return p; }
  static TOHC_DEV void fence() {}
};
#endif

enum { KIND_BOX = 0, KIND_HULL = 1, KIND_SPHERE = 2, KIND_CYLZ = 3 };

template <int G>
struct Cvx {
  static constexpr int VPL = 64 / G;
  int kind, cast, n;
  V3 dims;  // box: m_implicitShapeDimensions after the margin round trip (host computes it, engine.hip); sphere: (r, r, r)
  float x[VPL], y[VPL], z[VPL];
  Xf t01;   // cast: T_0_1 = T_w_0^-1 * T_w_1
};

// G = 1: the hull stays in memory (64 pairs per wavefront leave no registers for vertices); a support query is a scan
// The table it scans is BLOCKED: per hull, blocks of 8 vertices stored as x[8] y[8] z[8] (96 bytes, six 16-byte loads),
// the last block filled up with copies of the hull's last vertex -- a copy never wins the strict '>' of the scan, so no
// lane of the compare needs a mask, and two vertices' products and sums go through one packed instruction each.
template <>
struct Cvx<1> {
  int kind, cast, n;
  V3 dims;
  const float *vb;   // the hull's first block
  Xf t01;
};
#define TOHC_VBLOCK 24   // floats per block of 8 vertices
// the blocked table of a scene's hulls (host): voff8[shape] = first block of the shape's vertices
template <class VecF, class VecI>
inline void blockVertices(const float *vx, const float *vy, const float *vz, int n_shapes, const int *kind, const int *voff, const int *vcnt, VecF &v8,
                          VecI &voff8) {
  v8.clear(); voff8.assign(n_shapes, 0);
  for (int sh = 0; sh < n_shapes; ++sh) {
    if (kind[sh] != 1 /* KIND_HULL */ || vcnt[sh] <= 0) continue;
    voff8[sh] = (int)(v8.size() / TOHC_VBLOCK);
    for (int i0 = 0; i0 < vcnt[sh]; i0 += 8) {
      const size_t b = v8.size();
      v8.resize(b + TOHC_VBLOCK);
      for (int k = 0; k < 8; ++k) {
        const int i = voff[sh] + (i0 + k < vcnt[sh] ? i0 + k : vcnt[sh] - 1);
        v8[b + k] = vx[i]; v8[b + 8 + k] = vy[i]; v8[b + 16 + k] = vz[i];
      }
    }
  }
  v8.resize(v8.size() + TOHC_VBLOCK, 0.0f);   // (never empty)
}

template <int G>
TOHC_DEV void cvxLoad(Cvx<G> &c, int kind, V3 dims, const float *vx, const float *vy, const float *vz, const F4 *v4, int voff, int n,
                      const float * = nullptr, int = 0) {
  c.kind = kind;
  c.cast = 0;
  c.n = n;
  c.dims = dims;
  const int gl = Grp<G>::lane();
#pragma unroll
  for (int s = 0; s < Cvx<G>::VPL; ++s) {
    int i = s * G + gl;
    bool in = (kind == KIND_HULL) && (i < n);
    if (G >= 16) {   // SoA: the group's lanes read consecutive floats
      c.x[s] = in ? vx[voff + i] : 0.0f;
      c.y[s] = in ? vy[voff + i] : 0.0f;
      c.z[s] = in ? vz[voff + i] : 0.0f;
    } else {         // small groups: one 16-byte record per vertex (G consecutive records = G*16 contiguous bytes)
      F4 p;
      p.x = p.y = p.z = p.w = 0.0f;
      if (in) p = v4[voff + i];
      c.x[s] = p.x; c.y[s] = p.y; c.z[s] = p.z;
    }
  }
}

template <>
TOHC_DEV void cvxLoad<1>(Cvx<1> &c, int kind, V3 dims, const float *, const float *, const float *, const F4 *, int, int n, const float *v8,
                         int voff8) {
  c.kind = kind;
  c.cast = 0;
  c.n = n;
  c.dims = dims;
  c.vb = v8 + (size_t)voff8 * TOHC_VBLOCK;
}

TOHC_DEV float fsel(float a, float b, float c) { return a >= 0 ? b : c; }  // btScalar.h:468-473
// CylinderLocalSupportZ [cvx:156-215 with upAxis 2]; he = (radius, radius, half height)
TOHC_DEV V3 cylZSupport(V3 he, V3 v) {
  const float radius = he.x, halfHeight = he.z;
  const float s = sqrtf(v.x * v.x + v.y * v.y);
  const float z = (double)v.z < 0.0 ? -halfHeight : halfHeight;
  if (s != 0.0f) {
    const float d = radius / s;
    return mk(v.x * d, v.y * d, z);
  }
  return mk(radius, 0.0f, z);
}

// btVector3::maxDot (btVector3.h:1003-1018): strict '>' from index 0 => the lowest index among equal maxima
template <int G>
TOHC_DEV V3 hullMaxDot(const Cvx<G> &c, V3 d) {
  const int gl = Grp<G>::lane();
  float best = -INFINITY;
  int bi = INT_MAX;
#pragma unroll
  for (int s = 0; s < Cvx<G>::VPL; ++s) {
    int i = s * G + gl;
#if defined(__HIPCC__)
    if (__builtin_amdgcn_ballot_w64(i < c.n) == 0) break;   // no hull of this wavefront has a vertex in this slot or a later one
#endif
    float dt = c.x[s] * d.x + c.y[s] * d.y + c.z[s] * d.z;
    if (i < c.n && dt > best) { best = dt; bi = i; }
  }
  Grp<G>::argmax(best, bi);
  if (bi == INT_MAX) bi = 0;
  const int slot = bi / G, src = bi % G;
  float sx = c.x[0], sy = c.y[0], sz = c.z[0];
#pragma unroll
  for (int s = 1; s < Cvx<G>::VPL; ++s)
    if (s == slot) { sx = c.x[s]; sy = c.y[s]; sz = c.z[s]; }
  return mk(Grp<G>::bcast(sx, src), Grp<G>::bcast(sy, src), Grp<G>::bcast(sz, src));
}

// The scan takes a block of 8 vertices per step (six 16-byte loads issued together: with one load per vertex the loop
// would be a chain of L2 round trips, ~300 cycles each).  Products and sums are Bullet's, unfused, in its order
// ((x dx + y dy) + z dz); two vertices share each packed multiply / add.
#if defined(__HIPCC__)
typedef float Pk2 __attribute__((ext_vector_type(2)));
#endif
struct VBlock { F4 xa, xb, ya, yb, za, zb; };   // one block of the table in registers: x[8] y[8] z[8]
TOHC_DEV VBlock ldBlock(const float *vb, int i0) {
  const F4 *b = reinterpret_cast<const F4 *>(vb + 3 * i0);
  VBlock o;
  o.xa = b[0]; o.xb = b[1]; o.ya = b[2]; o.yb = b[3]; o.za = b[4]; o.zb = b[5];
  return o;
}
// the scan's step over one block: (best, bv) = the running maximum of dot(d, vertex) and its vertex
TOHC_DEV void scanBlock(const VBlock &B, V3 d, float &best, V3 &bv) {
  const float xs[8] = {B.xa.x, B.xa.y, B.xa.z, B.xa.w, B.xb.x, B.xb.y, B.xb.z, B.xb.w};
  const float ys[8] = {B.ya.x, B.ya.y, B.ya.z, B.ya.w, B.yb.x, B.yb.y, B.yb.z, B.yb.w};
  const float zs[8] = {B.za.x, B.za.y, B.za.z, B.za.w, B.zb.x, B.zb.y, B.zb.z, B.zb.w};
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
#if defined(__HIPCC__)
    const Pk2 X = {xs[k], xs[k + 1]}, Y = {ys[k], ys[k + 1]}, Z = {zs[k], zs[k + 1]};
    const Pk2 dt = (X * d.x + Y * d.y) + Z * d.z;
    const float dt0 = dt.x, dt1 = dt.y;
#else
    const float dt0 = xs[k] * d.x + ys[k] * d.y + zs[k] * d.z, dt1 = xs[k + 1] * d.x + ys[k + 1] * d.y + zs[k + 1] * d.z;
#endif
    if (dt0 > best) { best = dt0; bv = mk(xs[k], ys[k], zs[k]); }
    if (dt1 > best) { best = dt1; bv = mk(xs[k + 1], ys[k + 1], zs[k + 1]); }
  }
}
TOHC_DEV V3 scanResult(const Cvx<1> &c, float best, V3 bv) {
  if (!(best > -INFINITY) && c.n > 0) bv = mk(c.vb[0], c.vb[8], c.vb[16]);  // all-NaN dots: index 0, like the scan
  return bv;
}
template <>
TOHC_DEV V3 hullMaxDot<1>(const Cvx<1> &c, V3 d) {
  float best = -INFINITY;
  V3 bv = mk(0, 0, 0);
  for (int i0 = 0; i0 < c.n; i0 += 8) {
    const VBlock b = ldBlock(c.vb, i0);
    scanBlock(b, d, best, bv);
  }
  return scanResult(c, best, bv);
}
// The same scan for TWO directions in one pass over the table (the swept shape's support asks the same hull along d and
// along d in the second pose's frame): each block is loaded once.  Per direction the arithmetic and the order are hullMaxDot's.
TOHC_DEV void hullMaxDot2(const Cvx<1> &c, V3 d0, V3 d1, V3 &v0, V3 &v1) {
  float best0 = -INFINITY, best1 = -INFINITY;
  V3 bv0 = mk(0, 0, 0), bv1 = mk(0, 0, 0);
  for (int i0 = 0; i0 < c.n; i0 += 8) {
    const VBlock b = ldBlock(c.vb, i0);
    scanBlock(b, d0, best0, bv0);
    scanBlock(b, d1, best1, bv1);
  }
  v0 = scanResult(c, best0, bv0);
  v1 = scanResult(c, best1, bv1);
}
// Two HULLS at once -- the two operands of a Minkowski support query are independent scans, each a chain of L2 round
// trips (a block's six loads, then its compares): walking both tables in one loop keeps twice the loads in flight.  A
// hull with fewer blocks than the other re-reads its last block (a copy never wins the strict '>').  two: the hull is
// swept, its second direction is scanned in the same pass (hullMaxDot2).
TOHC_DEV void hullMaxDotPair(const Cvx<1> &A, V3 a0, bool twoA, V3 a1, const Cvx<1> &B, V3 b0, bool twoB, V3 b1, V3 &vA0, V3 &vA1, V3 &vB0, V3 &vB1) {
  float bestA0 = -INFINITY, bestA1 = -INFINITY, bestB0 = -INFINITY, bestB1 = -INFINITY;
  V3 pA0 = mk(0, 0, 0), pA1 = mk(0, 0, 0), pB0 = mk(0, 0, 0), pB1 = mk(0, 0, 0);
  const int nA = A.n > 0 ? ((A.n - 1) & ~7) : 0, nB = B.n > 0 ? ((B.n - 1) & ~7) : 0;   // first vertex of each hull's last block
  const int nMax = nA > nB ? nA : nB;
  for (int i0 = 0; i0 <= nMax; i0 += 8) {
    const VBlock ba = ldBlock(A.vb, i0 < nA ? i0 : nA);
    const VBlock bb = ldBlock(B.vb, i0 < nB ? i0 : nB);
    scanBlock(ba, a0, bestA0, pA0);
    if (twoA) scanBlock(ba, a1, bestA1, pA1);
    scanBlock(bb, b0, bestB0, pB0);
    if (twoB) scanBlock(bb, b1, bestB1, pB1);
  }
  vA0 = scanResult(A, bestA0, pA0); vA1 = scanResult(A, bestA1, pA1);
  vB0 = scanResult(B, bestB0, pB0); vB1 = scanResult(B, bestB1, pB1);
}
// vertex i of a hull, the same value in every lane of the group
template <int G>
TOHC_DEV V3 cvxVertex(const Cvx<G> &c, int i) {
  const int slot = i / G, src = i % G;
  float sx = c.x[0], sy = c.y[0], sz = c.z[0];
#pragma unroll
  for (int s = 1; s < Cvx<G>::VPL; ++s)
    if (s == slot) { sx = c.x[s]; sy = c.y[s]; sz = c.z[s]; }
  return mk(Grp<G>::bcast(sx, src), Grp<G>::bcast(sy, src), Grp<G>::bcast(sz, src));
}
template <>
TOHC_DEV V3 cvxVertex<1>(const Cvx<1> &c, int i) {
  const float *b = c.vb + (i >> 3) * TOHC_VBLOCK + (i & 7);
  return mk(b[0], b[8], b[16]);
}

// localGetSupportVertexWithoutMarginNonVirtual for box / hull [cvx:120-296]
// getMargin() of the plain shape: a sphere's radius (getRadius() = m_implicitShapeDimensions.x * m_localScaling.x,
// btSphereShape.h:62-67), 0 for everything else (MARGIN, bcc:23).  Derived on demand: no register of its own.
template <int G>
TOHC_DEV float shapeMargin(const Cvx<G> &c) { return c.kind == KIND_SPHERE ? c.dims.x * 1.0f : 0.0f; }
TOHC_DEV V3 normalizedOrMinus1(V3 d) {  // [cvx:298-306], btSphereShape.cpp:43-48
  if (len2(d) < (FLT_EPSILON * FLT_EPSILON)) d = mk(-1, -1, -1);
  return vdiv(d, len(d));
}
template <int G>
TOHC_DEV V3 supPlainNV(const Cvx<G> &c, V3 d) {
  if (c.kind == KIND_SPHERE) return mk(0, 0, 0);   // [cvx:125-128]
  if (c.kind == KIND_BOX) return mk(fsel(-d.x, -c.dims.x, c.dims.x), fsel(-d.y, -c.dims.y, c.dims.y), fsel(-d.z, -c.dims.z, c.dims.z));
  if (c.kind == KIND_CYLZ) return cylZSupport(c.dims, d);
  return hullMaxDot<G>(c, d);
}
// the VIRTUAL localGetSupportingVertex CastHullShape calls on its inner shape: box tie rule d >= 0 -> +h (btBoxShape.h:50-59)
template <int G>
TOHC_DEV V3 supPlainVirtual(const Cvx<G> &c, V3 d) {
  if (c.kind == KIND_SPHERE) return normalizedOrMinus1(d) * (c.dims.x * 1.0f);   // btSphereShape::localGetSupportingVertex (btSphereShape.cpp:38-53)
  if (c.kind == KIND_BOX) return mk(fsel(d.x, c.dims.x, -c.dims.x), fsel(d.y, c.dims.y, -c.dims.y), fsel(d.z, c.dims.z, -c.dims.z));
  if (c.kind == KIND_CYLZ) return cylZSupport(c.dims, d);   // btCylinderShape's virtual support + margin 0
  return hullMaxDot<G>(c, d);
}
// the swept shape's support from its two poses' supports (sv1 still in the second pose's frame) [bcc:826-830]
template <int G>
TOHC_DEV V3 castPick(const Cvx<G> &c, V3 d, V3 sv0, V3 sv1l) {
  const V3 sv1 = xfApply(c.t01, sv1l);
  return (dot(d, sv0) > dot(d, sv1)) ? sv0 : sv1;
}
template <int G>
TOHC_DEV V3 supNV(const Cvx<G> &c, V3 d) {
  if (!c.cast) return supPlainNV<G>(c, d);
  if constexpr (G == 1) {
    if (c.kind == KIND_HULL) {   // one pass over the table for both poses' directions
      V3 sv0, sv1;
      hullMaxDot2(c, d, mulVM(d, c.t01.b), sv0, sv1);
      return castPick<G>(c, d, sv0, sv1);
    }
  }
  V3 sv0 = supPlainVirtual<G>(c, d);  // CastHullShape::localGetSupportingVertex [bcc:826-830]
  return castPick<G>(c, d, sv0, supPlainVirtual<G>(c, mulVM(d, c.t01.b)));
}
// supNV of both operands of a Minkowski support query.  One lane per pair and two hulls: ONE loop over both tables
// (hullMaxDotPair); everything else: the two queries one after the other.  Values are supNV's.
template <int G>
TOHC_DEV void supPairNV(const Cvx<G> &A, V3 dA, const Cvx<G> &B, V3 dB, V3 &pA, V3 &qB) {
  if constexpr (G == 1) {
    if (A.kind == KIND_HULL && B.kind == KIND_HULL) {
      const bool twoA = A.cast != 0, twoB = B.cast != 0;
      V3 a0, a1, b0, b1;
      hullMaxDotPair(A, dA, twoA, twoA ? mulVM(dA, A.t01.b) : dA, B, dB, twoB, twoB ? mulVM(dB, B.t01.b) : dB, a0, a1, b0, b1);
      pA = twoA ? castPick<G>(A, dA, a0, a1) : a0;
      qB = twoB ? castPick<G>(B, dB, b0, b1) : b0;
      return;
    }
  }
  pA = supNV<G>(A, dA);
  qB = supNV<G>(B, dB);
}
// getMarginNonVirtual() of a GJK operand: the swept shape's is 0 (CastHullShape::getMargin, bcc:870)
template <int G>
TOHC_DEV float cvxMargin(const Cvx<G> &c) { return c.cast ? 0.0f : shapeMargin<G>(c); }
// localGetSupportVertexNonVirtual [cvx:298-308]
template <int G>
TOHC_DEV V3 supWithMarginNV(const Cvx<G> &c, V3 d) {
  const V3 n = normalizedOrMinus1(d);
  const float m = cvxMargin<G>(c);
  V3 s = supNV<G>(c, n);
  if (m != 0.0f) s = s + n * m;
  return s;
}

// ---------------------------------------------------------------------------------------------------------------
// The engine's own pre-test (k_broadphase / k_exact; nothing of Bullet's) needs no support POINT, only how far a shape
// reaches along a direction: max over the shape of dot(d, p), local frame.  That is a third of the instructions of the
// support query per hull vertex (mul, fma, fma, max against three products, two sums, a compare and four selects), and
// the value may round differently from dot(d, supNV(d)): the pre-test's guard band (1 % of d + 0.1 mm) is there for that.
// The shape is supNV's: without its margin when plain, the swept hull of both poses (a sphere with its radius) when cast.
// ---------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
// A pointer that is the same in every lane of the wavefront, as one the compiler KNOWS is uniform and read-only: loads
// through it are scalar loads (SGPR results, the scalar cache), not 64 copies of one value through the vector memory
// path -- a wave-uniform 16-byte vector load costs that path the same 1 KB of register writes as a gather.  Only for
// tables written before the kernel started (the scalar cache is not coherent with this kernel's own stores).
#define TOHC_CONST_AS __attribute__((address_space(4)))
template <class T>
TOHC_DEV const TOHC_CONST_AS T *uniformPtr(const T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  return (const TOHC_CONST_AS T *)(((unsigned long long)hi << 32) | lo);
}
// UNIFORM: the shape is the same in every lane (k_broadphase's query child): its vertices come through scalar loads
template <bool UNIFORM>
TOHC_DEV float plainReach(const Cvx<1> &c, V3 d, bool withRadius) {
  if (c.kind == KIND_SPHERE) return withRadius ? c.dims.x * sqrtf(len2(d)) : 0.0f;
  if (c.kind == KIND_BOX) return fmaf(fabsf(d.x), c.dims.x, fmaf(fabsf(d.y), c.dims.y, fabsf(d.z) * c.dims.z));
  if (c.kind == KIND_CYLZ) return fmaf(c.dims.x, sqrtf(fmaf(d.x, d.x, d.y * d.y)), fabsf(d.z) * c.dims.z);
  float best = -INFINITY;
  if (UNIFORM) {
    const int n = __builtin_amdgcn_readfirstlane(c.n);
    const TOHC_CONST_AS float *b = uniformPtr(c.vb);
    for (int i0 = 0; i0 < n; i0 += 8, b += TOHC_VBLOCK) {
#pragma unroll
      for (int k = 0; k < 8; ++k) best = fmaxf(best, fmaf(b[k], d.x, fmaf(b[8 + k], d.y, b[16 + k] * d.z)));
    }
    return best;
  }
  for (int i0 = 0; i0 < c.n; i0 += 8) {
    const F4 *b = reinterpret_cast<const F4 *>(c.vb + 3 * i0);
    const F4 xa = b[0], xb = b[1], ya = b[2], yb = b[3], za = b[4], zb = b[5];
    const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
    const float ys[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
    const float zs[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) best = fmaxf(best, fmaf(xs[k], d.x, fmaf(ys[k], d.y, zs[k] * d.z)));
  }
  return best;
}
template <bool UNIFORM = false>
TOHC_DEV float reachNV(const Cvx<1> &c, V3 d) {
  if (!c.cast) return plainReach<UNIFORM>(c, d, false);
  const float e0 = plainReach<UNIFORM>(c, d, true);
  const float e1 = plainReach<UNIFORM>(c, mulVM(d, c.t01.b), true) + dot(d, c.t01.o);
  return fmaxf(e0, e1);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// btVoronoiSimplexSolver [vss]; arrays are only ever indexed with compile-time constants (select chains), so they
// stay in registers
// ---------------------------------------------------------------------------------------------------------------
struct SubRes {
  V3 closest;
  bool used[4];
  float bc[4];
  bool degenerate;
};
struct Simplex {
  int n;
  V3 W[4], P[4], Q[4];
  V3 cP1, cP2, cV, lastW;
  bool cachedValid, needsUpdate;
  SubRes bc;
};

// value selects, not conditional stores: a store through a phi of addresses keeps the whole simplex in scratch memory
TOHC_DEV V3 sel3(bool c, V3 a, V3 b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
TOHC_DEV V3 pick(const V3 *a, int i) {
  V3 r = a[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) r = sel3(k == i, a[k], r);
  return r;
}
TOHC_DEV void put(V3 *a, int i, V3 v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = sel3(k == i, v, a[k]);
}
TOHC_DEV void subResReset(SubRes &r) {
  r.degenerate = false;
  r.bc[0] = r.bc[1] = r.bc[2] = r.bc[3] = 0;
  r.used[0] = r.used[1] = r.used[2] = r.used[3] = false;
}
TOHC_DEV void setBC(SubRes &r, float a, float b, float c, float d) { r.bc[0] = a; r.bc[1] = b; r.bc[2] = c; r.bc[3] = d; }
TOHC_DEV bool subResValid(const SubRes &r) { return r.bc[0] >= 0 && r.bc[1] >= 0 && r.bc[2] >= 0 && r.bc[3] >= 0; }
TOHC_DEV void vssReset(Simplex &s) {
  s.cachedValid = false;
  s.n = 0;
  s.needsUpdate = true;
  s.lastW = mk(1e18f, 1e18f, 1e18f);
  subResReset(s.bc);
#if !defined(__HIPCC__)
  for (int k = 0; k < 4; ++k) s.W[k] = s.P[k] = s.Q[k] = mk(0, 0, 0);  // host build: keep sanitizers quiet
  s.cP1 = s.cP2 = s.cV = mk(0, 0, 0);
#endif
}
template <int I>
TOHC_DEV void vssRemove(Simplex &s) {
  s.n--;
  s.W[I] = pick(s.W, s.n);
  s.P[I] = pick(s.P, s.n);
  s.Q[I] = pick(s.Q, s.n);
}
TOHC_DEV void vssReduce(Simplex &s, const bool *used) {  // [vss:45-59]
  if (s.n >= 4 && !used[3]) vssRemove<3>(s);
  if (s.n >= 3 && !used[2]) vssRemove<2>(s);
  if (s.n >= 2 && !used[1]) vssRemove<1>(s);
  if (s.n >= 1 && !used[0]) vssRemove<0>(s);
}
TOHC_DEV void vssAdd(Simplex &s, V3 w, V3 p, V3 q) {
  s.lastW = w;
  s.needsUpdate = true;
  put(s.W, s.n, w);
  put(s.P, s.n, p);
  put(s.Q, s.n, q);
  s.n++;
}
TOHC_DEV bool vssInSimplex(const Simplex &s, V3 w) {  // [vss:283-305]; BT_USE_EQUAL_VERTEX_THRESHOLD is defined (vss.h:28)
  bool found = false;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < s.n && len2(w - s.W[i]) <= 0.0001f) found = true;
  if (veq(w, s.lastW)) return true;
  return found;
}

TOHC_DEV void closestPtTriangle(V3 p, V3 a, V3 b, V3 c, SubRes &r) {  // [vss:330-423]
  r.used[0] = r.used[1] = r.used[2] = r.used[3] = false;
  V3 ab = b - a, ac = c - a, ap = p - a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) { r.closest = a; r.used[0] = true; setBC(r, 1, 0, 0, 0); return; }
  V3 bp = p - b;
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) { r.closest = b; r.used[1] = true; setBC(r, 0, 1, 0, 0); return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    r.closest = a + ab * v;
    r.used[0] = r.used[1] = true;
    setBC(r, 1 - v, v, 0, 0);
    return;
  }
  V3 cp = p - c;
  float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) { r.closest = c; r.used[2] = true; setBC(r, 0, 0, 1, 0); return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    r.closest = a + ac * w;
    r.used[0] = r.used[2] = true;
    setBC(r, 1 - w, 0, w, 0);
    return;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    r.closest = b + (c - b) * w;
    r.used[1] = r.used[2] = true;
    setBC(r, 0, 1 - w, w, 0);
    return;
  }
  float denom = 1.0f / (va + vb + vc);
  float v = vb * denom, w = vc * denom;
  r.closest = (a + ab * v) + ac * w;
  r.used[0] = r.used[1] = r.used[2] = true;
  setBC(r, 1 - v - w, v, w, 0);
}

TOHC_DEV int pointOutsideOfPlane(V3 p, V3 a, V3 b, V3 c, V3 d) {  // [vss:430-454]
  V3 normal = cross(b - a, c - a);
  float signp = dot(p - a, normal);
  float signd = dot(d - a, normal);
  if (signd * signd < (1e-4f * 1e-4f)) return -1;
  return signp * signd < 0.0f;
}

TOHC_DEV bool closestPtTetrahedron(V3 p, V3 a, V3 b, V3 c, V3 d, SubRes &fr) {  // [vss:457-608]
  SubRes tr;
  fr.closest = p;
  fr.used[0] = fr.used[1] = fr.used[2] = fr.used[3] = true;
  int oABC = pointOutsideOfPlane(p, a, b, c, d);
  int oACD = pointOutsideOfPlane(p, a, c, d, b);
  int oADB = pointOutsideOfPlane(p, a, d, b, c);
  int oBDC = pointOutsideOfPlane(p, b, d, c, a);
  if (oABC < 0 || oACD < 0 || oADB < 0 || oBDC < 0) { fr.degenerate = true; return false; }
  if (!oABC && !oACD && !oADB && !oBDC) return false;
  float best = FLT_MAX;
  if (oABC) {
    closestPtTriangle(p, a, b, c, tr);
    V3 q = tr.closest;
    float sq = dot(q - p, q - p);
    if (sq < best) {
      best = sq; fr.closest = q;
      fr.used[0] = tr.used[0]; fr.used[1] = tr.used[1]; fr.used[2] = tr.used[2]; fr.used[3] = false;
      setBC(fr, tr.bc[0], tr.bc[1], tr.bc[2], 0);
    }
  }
  if (oACD) {
    closestPtTriangle(p, a, c, d, tr);
    V3 q = tr.closest;
    float sq = dot(q - p, q - p);
    if (sq < best) {
      best = sq; fr.closest = q;
      fr.used[0] = tr.used[0]; fr.used[1] = false; fr.used[2] = tr.used[1]; fr.used[3] = tr.used[2];
      setBC(fr, tr.bc[0], 0, tr.bc[1], tr.bc[2]);
    }
  }
  if (oADB) {
    closestPtTriangle(p, a, d, b, tr);
    V3 q = tr.closest;
    float sq = dot(q - p, q - p);
    if (sq < best) {
      best = sq; fr.closest = q;
      fr.used[0] = tr.used[0]; fr.used[1] = tr.used[2]; fr.used[2] = false; fr.used[3] = tr.used[1];
      setBC(fr, tr.bc[0], tr.bc[2], 0, tr.bc[1]);
    }
  }
  if (oBDC) {
    closestPtTriangle(p, b, d, c, tr);
    V3 q = tr.closest;
    float sq = dot(q - p, q - p);
    if (sq < best) {
      best = sq; fr.closest = q;
      fr.used[0] = false; fr.used[1] = tr.used[0]; fr.used[2] = tr.used[2]; fr.used[3] = tr.used[1];
      setBC(fr, 0, tr.bc[0], tr.bc[2], tr.bc[1]);
    }
  }
  return true;
}

TOHC_DEV bool vssUpdate(Simplex &s) {  // updateClosestVectorAndPoints [vss:90-240]
  if (!s.needsUpdate) return s.cachedValid;
  subResReset(s.bc);
  s.needsUpdate = false;
  const V3 O = mk(0, 0, 0);
  switch (s.n) {
    case 0: s.cachedValid = false; break;
    case 1:
      s.cP1 = s.P[0];
      s.cP2 = s.Q[0];
      s.cV = s.cP1 - s.cP2;
      subResReset(s.bc);
      setBC(s.bc, 1, 0, 0, 0);
      s.cachedValid = subResValid(s.bc);
      break;
    case 2: {
      V3 from = s.W[0], to = s.W[1];
      V3 diff = O - from;
      V3 v = to - from;
      float t = dot(v, diff);
      if (t > 0) {
        float dotVV = dot(v, v);
        if (t < dotVV) {
          t /= dotVV;
          diff = diff - v * t;
          s.bc.used[0] = true; s.bc.used[1] = true;
        } else {
          t = 1;
          diff = diff - v;
          s.bc.used[1] = true;
        }
      } else {
        t = 0;
        s.bc.used[0] = true;
      }
      setBC(s.bc, 1 - t, t, 0, 0);
      s.cP1 = s.P[0] + (s.P[1] - s.P[0]) * t;
      s.cP2 = s.Q[0] + (s.Q[1] - s.Q[0]) * t;
      s.cV = s.cP1 - s.cP2;
      vssReduce(s, s.bc.used);
      s.cachedValid = subResValid(s.bc);
      break;
    }
    case 3: {
      closestPtTriangle(O, s.W[0], s.W[1], s.W[2], s.bc);
      const float *b = s.bc.bc;
      s.cP1 = (s.P[0] * b[0] + s.P[1] * b[1]) + s.P[2] * b[2];
      s.cP2 = (s.Q[0] * b[0] + s.Q[1] * b[1]) + s.Q[2] * b[2];
      s.cV = s.cP1 - s.cP2;
      vssReduce(s, s.bc.used);
      s.cachedValid = subResValid(s.bc);
      break;
    }
    case 4: {
      bool sep = closestPtTetrahedron(O, s.W[0], s.W[1], s.W[2], s.W[3], s.bc);
      if (sep) {
        const float *b = s.bc.bc;
        s.cP1 = ((s.P[0] * b[0] + s.P[1] * b[1]) + s.P[2] * b[2]) + s.P[3] * b[3];
        s.cP2 = ((s.Q[0] * b[0] + s.Q[1] * b[1]) + s.Q[2] * b[2]) + s.Q[3] * b[3];
        s.cV = s.cP1 - s.cP2;
        vssReduce(s, s.bc.used);
      } else {
        if (s.bc.degenerate) {
          s.cachedValid = false;
        } else {
          s.cachedValid = true;
          s.cV = O;
        }
        break;
      }
      s.cachedValid = subResValid(s.bc);
      break;
    }
    default: s.cachedValid = false;
  }
  return s.cachedValid;
}

// ---------------------------------------------------------------------------------------------------------------
// GJK main loop [gjk:82-332].  Returns the state the EPA stage needs.
// ---------------------------------------------------------------------------------------------------------------
struct GjkState {
  bool isValid, needEpa;
  float distance;
  V3 normalInB, pointOnB;   // pointOnB in the midpoint-shifted frame
  V3 offset;
  int iters;
  float mA, mB;   // the operands' margins (0 except for plain spheres)
};

// The loop of getClosestPointsNonVirtual as a resumable object: init(), step() until it returns true, finish().
// k_gjk keeps one of these per group and refills a group with its next pair as soon as its loop ends, so the groups
// of a wavefront never wait for the slowest pair (iteration counts range from 1 to ~8, two thirds end after two).
template <int G>
struct GjkRun {
  Simplex sx;
  V3 sep, offset, oA, oB;
  float squaredDistance;
  int curIter, degenerate;
  bool checkSimplex;

  TOHC_DEV void init(const Xf &tA, const Xf &tB) {
    offset = (tA.o + tB.o) * 0.5f;
    oA = tA.o - offset;
    oB = tB.o - offset;
    curIter = 0;
    sep = mk(0, 1, 0);
    checkSimplex = false;
    degenerate = 0;
    squaredDistance = 1e18f;
    vssReset(sx);
  }
  // one pass of the for(;;) body [gjk:137-302]; true = the loop has ended
  TOHC_DEV bool step(const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, float maxDist2, int maxIter) {
    V3 sepA = mulVM(-sep, tA.b);
    V3 sepB = mulVM(sep, tB.b);
    V3 pInA, qInB;
    supPairNV<G>(A, sepA, B, sepB, pInA, qInB);
    V3 pW = mulMV(tA.b, pInA) + oA;
    V3 qW = mulMV(tB.b, qInB) + oB;
    V3 w = pW - qW;
    float delta = dot(sep, w);
    if ((delta > 0.0f) && (delta * delta > squaredDistance * maxDist2)) { degenerate = 10; checkSimplex = true; return true; }
    if (vssInSimplex(sx, w)) { degenerate = 1; checkSimplex = true; return true; }
    float f0 = squaredDistance - delta;
    float f1 = squaredDistance * 1.0e-6f;
    if (f0 <= f1) { degenerate = (f0 <= 0.0f) ? 2 : 11; checkSimplex = true; return true; }
    vssAdd(sx, w, pW, qW);
    bool ok = vssUpdate(sx);
    V3 newSep = sx.cV;
    if (!ok) { degenerate = 3; checkSimplex = true; return true; }
    if (len2(newSep) < 1.0e-6f) { sep = newSep; degenerate = 6; checkSimplex = true; return true; }
    float prev = squaredDistance;
    squaredDistance = len2(newSep);
    if (prev - squaredDistance <= FLT_EPSILON * prev) { checkSimplex = true; degenerate = 12; return true; }
    sep = newSep;
    if (curIter++ > maxIter) return true;
    if (sx.n == 4) { degenerate = 13; return true; }
    return false;
  }
  // [gjk:304-335]; mA / mB: the operands' margins (cvxMargin)
  TOHC_DEV void finish(GjkState &st, float mA, float mB) {
    float distance = 0;
    V3 normalInB = mk(0, 0, 0), pointOnB = mk(0, 0, 0);
    bool isValid = false;
    if (checkSimplex) {
      vssUpdate(sx);
      pointOnB = sx.cP2;
      normalInB = sep;
      float lenSqr = len2(sep);
      if ((double)lenSqr < 0.0001) degenerate = 5;
      if (lenSqr > FLT_EPSILON * FLT_EPSILON) {
        float rlen = 1.0f / sqrtf(lenSqr);
        normalInB = normalInB * rlen;
        if (mA != 0.0f || mB != 0.0f) {   // [gjk:318-325]; with zero margins these are exact no-ops
          const float s = sqrtf(squaredDistance);
          pointOnB = pointOnB + sep * (mB / s);
          distance = (1.0f / rlen) - (mA + mB);
        } else {
          distance = (1.0f / rlen);
        }
        isValid = true;
      }
    }
    bool catchDeg = degenerate && ((double)(distance + (mA + mB)) < 0.01);
    st.isValid = isValid;
    st.needEpa = (!isValid || catchDeg);
    st.distance = distance;
    st.normalInB = normalInB;
    st.pointOnB = pointOnB;
    st.offset = offset;
    st.iters = curIter;
    st.mA = mA; st.mB = mB;
  }
};

// Conservative separating-axis pre-test (no counterpart in the reference; it only removes pairs the reference would
// evaluate and then drop): along ANY world direction `dir` the support points pA (of A, towards -dir) and qB (of B,
// towards +dir) bound the distance from below by dot(dir, pA - qB) / |dir|.  When that bound exceeds `lim` (the
// contact distance plus a guard band far above fp32 rounding of these few operations) no contact with distance <= d
// exists, so the collector's filter (bullet_collision_checker.cpp:401) would discard whatever GJK returns.
template <int G>
TOHC_DEV bool sepAxisBeyond(const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, V3 dir, float lim, V3 &w) {
  V3 pA = xfApply(tA, supNV<G>(A, mulVM(-dir, tA.b)));
  V3 qB = xfApply(tB, supNV<G>(B, mulVM(dir, tB.b)));
  w = pA - qB;
  float delta = dot(dir, w);
  const float limm = lim + (cvxMargin<G>(A) + cvxMargin<G>(B));   // the supports are those of the cores: a plain sphere reaches its radius further
  return (delta > 0.0f) && (delta * delta > (limm * limm) * len2(dir));
}

template <int G>
TOHC_DEV void gjkMain(const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, float maxDist2, GjkState &st, int maxIter = 1000) {
  GjkRun<G> r;
  r.init(tA, tB);
  while (!r.step(A, B, tA, tB, maxDist2, maxIter)) {}
  r.finish(st, cvxMargin<G>(A), cvxMargin<G>(B));
}

// ---------------------------------------------------------------------------------------------------------------
// GJK2 + EPA [epa].  All pools live in EpaMem (LDS on the device), indexed by small integers instead of pointers;
// the recursions of EncloseOrigin and expand are unrolled onto explicit stacks that visit faces in the same order.
// ---------------------------------------------------------------------------------------------------------------
#define TOHC_EPA_MAX_VERTICES 64   // the reference's limits (EPA_MAX_VERTICES, EPA_MAX_FACES): part of the results
#define TOHC_EPA_MAX_FACES 128
// The small pools k_tail_epa tries first: a run that stays inside them is, step for step, the run on the full pools
// (the lowest free face slot is the same slot in both), and one that would leave them is started again on the full
// ones.  Measured on config 4 (74 K runs): 99.98 % add <= 16 support vertices and none more than 24; 99.9 % never touch
// a face slot >= 32 and one reaches slot 50.  The runs that leave the pools are the longest ones, and repeating one costs
// its whole length again at the end of the kernel, so the pools are as large as the occupancy the registers give
// (8 wavefronts per CU, 8 groups each: 160 KB / 64 = 2.5 KB) lets them be.
#ifndef TOHC_EPA_SMALL_V
#define TOHC_EPA_SMALL_V 23   // (23 / 48: 2544 B, eight of them in 160 KB / 8 blocks per CU)
#define TOHC_EPA_SMALL_F 48
#endif
#define TOHC_NIL 255
#ifndef TOHC_EPA_FAST_EXPAND
#define TOHC_EPA_FAST_EXPAND 1   // 0: every expansion through traverse(), the exact walk (tests compile the host build both ways)
#endif

// One record per access: a face is read as (normal, distance) = one 16-byte load, its three vertex ids + pass mark
// = one 4-byte load, its neighbours and their edge numbers = two 4-byte loads -- EPA is a chain of dependent LDS round
// trips, so what counts is how few of them a step needs, not how many bytes.  Which faces are on the hull / free is
// kept in two 128-bit masks in registers (Epa), not in linked lists: the list order matters to the reference only in
// findbest's tie rule, which the arrival stamp reproduces.
template <int NV, int NF>
struct alignas(16) EpaMemT {
  static constexpr int kNV = NV, kNF = NF;
  // support vertices: 0..3 = GJK2's m_store, 4.. = EPA's m_sv_store
  float svd[4 + NV][4];
  float svw[4 + NV][4];
  float fnd[NF][4];      // n.x n.y n.z d
  uint32_t fcp[NF];      // bytes: c[0] c[1] c[2] pass
  uint32_t ff[NF];       // bytes: f[0] f[1] f[2] -
  uint32_t fe[NF];       // bytes: e[0] e[1] e[2] -
  uint16_t fstamp[NF];   // order of arrival on the hull (the reference's hull list is newest-first)
  uint16_t stk[NF];      // expand() stack: face | edge << 8 | stage << 10
  uint16_t hz[NF];       // the horizon of one expansion, in visiting order: face | edge << 7 | new face's slot << 9
  // GJK2 simplices
  uint8_t sc[2][4];
  float sp[2][4];
  uint8_t freeL[4];
};
typedef EpaMemT<TOHC_EPA_MAX_VERTICES, TOHC_EPA_MAX_FACES> EpaMem;   // the reference's pools
typedef EpaMemT<TOHC_EPA_SMALL_V, TOHC_EPA_SMALL_F> EpaMemS;        // first attempt (k_tail_epa)

// the part of EpaMem that GJK2 alone touches: one per LANE in the lane-per-pair penetration kernel (k_pen)
struct G2Mem {
  float svd[4][3];
  float svw[4][3];
  uint8_t sc[2][4];
  float sp[2][4];
  uint8_t freeL[4];
};

TOHC_DEV V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
TOHC_DEV void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

template <int G>
struct Mink {
  const Cvx<G> *s0, *s1;
  M3 toshape1;
  Xf toshape0;
  bool margins;
};
template <int G>
TOHC_DEV V3 minkSupport0(const Mink<G> &m, V3 d) { return m.margins ? supWithMarginNV<G>(*m.s0, d) : supNV<G>(*m.s0, d); }
template <int G>
TOHC_DEV V3 minkSupport1(const Mink<G> &m, V3 d) {
  V3 dl = mulMV(m.toshape1, d);
  V3 s = m.margins ? supWithMarginNV<G>(*m.s1, dl) : supNV<G>(*m.s1, dl);
  return xfApply(m.toshape0, s);
}
template <int G>
TOHC_DEV V3 minkSupport(const Mink<G> &m, V3 d) {
  if constexpr (G == 1) {   // both operands' scans in one loop (supPairNV); the values are those of the two calls below
    const V3 d1 = mulMV(m.toshape1, -d);
    const V3 n0 = m.margins ? normalizedOrMinus1(d) : d, n1 = m.margins ? normalizedOrMinus1(d1) : d1;
    V3 s0, s1;
    supPairNV<G>(*m.s0, n0, *m.s1, n1, s0, s1);
    if (m.margins) {
      const float m0 = cvxMargin<G>(*m.s0), m1 = cvxMargin<G>(*m.s1);
      if (m0 != 0.0f) s0 = s0 + n0 * m0;
      if (m1 != 0.0f) s1 = s1 + n1 * m1;
    }
    return s0 - xfApply(m.toshape0, s1);
  }
  return minkSupport0<G>(m, d) - minkSupport1<G>(m, -d);
}

TOHC_DEV float det3(V3 a, V3 b, V3 c) {
  return (a.y * b.z * c.x + a.z * b.x * c.y - a.x * b.z * c.y - a.y * b.x * c.z + a.x * b.y * c.z - a.z * b.y * c.x);
}
TOHC_DEV float projOrigin2(V3 a, V3 b, float *w, unsigned &m) {  // [epa:393-407]
  V3 d = b - a;
  float l = len2(d);
  if (l > 0.0f) {
    float t = (l > 0 ? -dot(a, d) / l : 0);
    if (t >= 1) { w[0] = 0; w[1] = 1; m = 2; return len2(b); }
    else if (t <= 0) { w[0] = 1; w[1] = 0; m = 1; return len2(a); }
    else { w[0] = 1 - (w[1] = t); m = 3; return len2(a + d * t); }
  }
  return -1;
}
TOHC_DEV void wset(float *w, unsigned i, float v) {
#pragma unroll
  for (unsigned k = 0; k < 4; ++k)
    if (k == i) w[k] = v;
}
TOHC_DEV float projOrigin3(V3 a, V3 b, V3 c, float *w, unsigned &m) {  // [epa:408-453]
  const V3 vt[3] = {a, b, c};
  const V3 dl[3] = {a - b, b - c, c - a};
  V3 n = cross(dl[0], dl[1]);
  float l = len2(n);
  if (l > 0.0f) {
    float mindist = -1;
    float subw[2] = {0.f, 0.f};
    unsigned subm = 0;
#pragma unroll
    for (unsigned i = 0; i < 3; ++i) {
      const unsigned j = (i + 1) % 3;
      if (dot(vt[i], cross(dl[i], n)) > 0) {
        float subd = projOrigin2(vt[i], vt[j], subw, subm);
        if ((mindist < 0) || (subd < mindist)) {
          mindist = subd;
          m = ((subm & 1) ? 1u << i : 0) + ((subm & 2) ? 1u << j : 0);
          w[i] = subw[0];
          w[j] = subw[1];
          w[(j + 1) % 3] = 0;
        }
      }
    }
    if (mindist < 0) {
      float d = dot(a, n);
      float s = sqrtf(l);
      V3 p = n * (d / l);
      mindist = len2(p);
      m = 7;
      w[0] = len(cross(dl[1], b - p)) / s;
      w[1] = len(cross(dl[2], c - p)) / s;
      w[2] = 1 - (w[0] + w[1]);
    }
    return mindist;
  }
  return -1;
}
TOHC_DEV float projOrigin4(V3 a, V3 b, V3 c, V3 d, float *w, unsigned &m) {  // [epa:454-502]
  const V3 vt[4] = {a, b, c, d};
  const V3 dl[3] = {a - d, b - d, c - d};
  float vl = det3(dl[0], dl[1], dl[2]);
  bool ng = (vl * dot(a, cross(b - c, a - b))) <= 0;
  if (ng && (fabsf(vl) > 0.0f)) {
    float mindist = -1;
    float subw[3] = {0.f, 0.f, 0.f};
    unsigned subm = 0;
#pragma unroll
    for (unsigned i = 0; i < 3; ++i) {
      const unsigned j = (i + 1) % 3;
      float s = vl * dot(d, cross(dl[i], dl[j]));
      if (s > 0) {
        float subd = projOrigin3(vt[i], vt[j], d, subw, subm);
        if ((mindist < 0) || (subd < mindist)) {
          mindist = subd;
          m = ((subm & 1) ? 1u << i : 0) + ((subm & 2) ? 1u << j : 0) + ((subm & 4) ? 8 : 0);
          w[i] = subw[0];
          w[j] = subw[1];
          w[(j + 1) % 3] = 0;
          w[3] = subw[2];
        }
      }
    }
    if (mindist < 0) {
      mindist = 0;
      m = 15;
      w[0] = det3(c, b, d) / vl;
      w[1] = det3(a, c, d) / vl;
      w[2] = det3(b, a, d) / vl;
      w[3] = 1 - (w[0] + w[1] + w[2]);
    }
    return mindist;
  }
  return -1;
}

enum { G2_VALID, G2_INSIDE, G2_FAILED };
#if defined(TOHC_EMU_STATS)
struct EmuStats { long gjk2_calls, gjk2_iters, epa_calls, epa_iters, epa_faces, enclose_appends, hist_epa[72], status[12], inside, hist_g2[130]; long last_g2; double last_d0; };
inline EmuStats &emuStats() { static EmuStats s; return s; }
#define TOHC_STAT(x) x
#else
#define TOHC_STAT(x)
#endif

template <int G, class MEM>
struct Gjk2 {
  MEM *m;
  Mink<G> shape;
  V3 ray;
  float distance;
  unsigned rank0, rank1;  // m_simplices[0/1].rank; two scalars + selects so that the struct stays in registers
  unsigned nfree, current;
  int status;
  unsigned nsup = 0;   // support queries so far (work counter; not part of the algorithm)
  unsigned iters = 0;  // passes of evaluate()'s loop (k_tail_pen files a pair by it: the fewer, the longer its EPA run)

  TOHC_DEV void support(V3 d, unsigned sv) {  // getsupport [epa:372-376]
    ++nsup;
    V3 dn = vdiv(d, len(d));
    st3(m->svd[sv], dn);
    st3(m->svw[sv], minkSupport<G>(shape, dn));
  }
  TOHC_DEV V3 W(unsigned s, unsigned i) const { return ld3(m->svw[m->sc[s][i]]); }
  TOHC_DEV unsigned rk(unsigned s) const { return s ? rank1 : rank0; }
  TOHC_DEV void setRk(unsigned s, unsigned v) { rank0 = s ? rank0 : v; rank1 = s ? v : rank1; }
  TOHC_DEV void remove(unsigned s) {
    const unsigned r = rk(s) - 1;
    setRk(s, r);
    m->freeL[nfree++] = m->sc[s][r];
  }
  TOHC_DEV void append(unsigned s, V3 v) {
    const unsigned r = rk(s);
    m->sp[s][r] = 0;
    unsigned sv = m->freeL[--nfree];
    m->sc[s][r] = (uint8_t)sv;
    setRk(s, r + 1);
    support(v, sv);
  }

  TOHC_DEV int evaluate(const Mink<G> &shapearg, V3 guess) {  // [epa:183-305]
    unsigned iterations = 0;
    float sqdist = 0, alpha = 0;
    V3 lastw[4];
    unsigned clastw = 0;
    m->freeL[0] = 0; m->freeL[1] = 1; m->freeL[2] = 2; m->freeL[3] = 3;
    nfree = 4;
    current = 0;
    status = G2_VALID;
    shape = shapearg;
    distance = 0;
    rank0 = 0; rank1 = 0;
    ray = guess;
    float sqrl = len2(ray);
    append(0, sqrl > 0 ? -ray : mk(1, 0, 0));
    m->sp[0][0] = 1;
    ray = W(0, 0);
    sqdist = sqrl;
    lastw[0] = lastw[1] = lastw[2] = lastw[3] = ray;
    do {
      const unsigned next = 1 - current;
      const unsigned cs = current, ns = next;
      float rl = len(ray);
      if (rl < 0.0001f) { status = G2_INSIDE; break; }
      append(cs, -ray);
      V3 w = W(cs, rk(cs) - 1);
      bool found = false;
#pragma unroll
      for (unsigned i = 0; i < 4; ++i)
        if (!found && len2(w - lastw[i]) < 0.0001f) found = true;
      if (found) { remove(current); break; }
      else {
        clastw = (clastw + 1) & 3;
        put(lastw, (int)clastw, w);
      }
      float omega = dot(ray, w) / rl;
      alpha = omega > alpha ? omega : alpha;
      if (((rl - alpha) - (0.0001f * rl)) <= 0) { remove(current); break; }
      float weights[4] = {0, 0, 0, 0};
      unsigned mask = 0;
      switch (rk(cs)) {
        case 2: sqdist = projOrigin2(W(cs, 0), W(cs, 1), weights, mask); break;
        case 3: sqdist = projOrigin3(W(cs, 0), W(cs, 1), W(cs, 2), weights, mask); break;
        case 4: sqdist = projOrigin4(W(cs, 0), W(cs, 1), W(cs, 2), W(cs, 3), weights, mask); break;
      }
      if (sqdist >= 0) {
        unsigned rn = 0;
        ray = mk(0, 0, 0);
        current = next;
        for (unsigned i = 0, ni = rk(cs); i < ni; ++i) {
          float wi = weights[0];
#pragma unroll
          for (unsigned k = 1; k < 4; ++k)
            if (k == i) wi = weights[k];
          if (mask & (1u << i)) {
            m->sc[ns][rn] = m->sc[cs][i];
            m->sp[ns][rn] = wi;
            rn++;
            ray = ray + W(cs, i) * wi;
          } else {
            m->freeL[nfree++] = m->sc[cs][i];
          }
        }
        setRk(ns, rn);
        if (mask == 15) status = G2_INSIDE;
      } else {
        remove(current);
        break;
      }
      status = ((++iterations) < 128) ? status : G2_FAILED;
    } while (status == G2_VALID);
    iters = iterations;
    TOHC_STAT(emuStats().last_g2 = (long)iterations;)
    TOHC_STAT(emuStats().gjk2_calls++; emuStats().gjk2_iters += iterations; emuStats().hist_g2[iterations < 129 ? iterations : 129]++; if (status == G2_INSIDE) emuStats().inside++;)
    switch (status) {
      case G2_VALID: distance = len(ray); break;
      case G2_INSIDE: distance = 0; break;
      default: break;
    }
    return status;
  }

  // EncloseOrigin [epa:306-370], recursion unrolled: cnt[level] = next candidate to try at that depth
  TOHC_DEV bool encloseOrigin() {
    const unsigned s = current;
    int lvl = 0;
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    for (;;) {
      const unsigned r = rk(s);
      if (r == 4) {
        if (fabsf(det3(W(s, 0) - W(s, 3), W(s, 1) - W(s, 3), W(s, 2) - W(s, 3))) > 0) return true;
        if (lvl == 0) return false;
        // return false to the parent: it removes the vertex it appended and tries its next candidate
        remove(s);
        --lvl;
        continue;
      }
      int k = lvl == 0 ? cnt0 : lvl == 1 ? cnt1 : lvl == 2 ? cnt2 : cnt3;
      const int maxc = (r == 3) ? 2 : 6;
      if (k >= maxc) {
        if (lvl == 0) return false;
        remove(s);
        --lvl;
        continue;
      }
      if (lvl == 0) cnt0 = k + 1; else if (lvl == 1) cnt1 = k + 1; else if (lvl == 2) cnt2 = k + 1; else cnt3 = k + 1;
      V3 dir;
      bool ok = true;
      const int ax = k >> 1;
      V3 axis = mk(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
      if (r == 1) {
        dir = axis;
      } else if (r == 2) {
        V3 d = W(s, 1) - W(s, 0);
        dir = cross(d, axis);
        ok = len2(dir) > 0;
      } else {
        dir = cross(W(s, 1) - W(s, 0), W(s, 2) - W(s, 0));
        ok = len2(dir) > 0;
      }
      if (!ok) continue;
      if (k & 1) dir = -dir;
      append(s, dir);
      ++lvl;
      if (lvl == 1) cnt1 = 0; else if (lvl == 2) cnt2 = 0; else cnt3 = 0;
    }
  }
};

// optional in-kernel profile of EPA (k_epa built with TOHC_EPA_PROF): cycle counts per phase, summed over evaluations
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
struct EpaProf { unsigned long long t[8]; unsigned n[8]; unsigned maxface, nsv, nslow, nfail; };
#define TOHC_PROF_T0(x) unsigned long long x = clock64()
#define TOHC_PROF_ACC(p, k, x) do { (p).t[k] += clock64() - (x); (p).n[k]++; } while (0)
#else
struct EpaProf {};
#define TOHC_PROF_T0(x)
#define TOHC_PROF_ACC(p, k, x)
#endif

enum { E_VALID, E_TOUCHING, E_DEGENERATED, E_NONCONVEX, E_INVALIDHULL, E_OUTOFFACES, E_OUTOFVERTICES, E_ACCURACY, E_FALLBACK, E_FAILED };

template <int G, class MEM = EpaMem>
struct Epa {
  MEM *m;
  int status;
  bool exhausted;   // a pool smaller than the reference's ran out: the run says nothing, start again on the full pools
  unsigned limV, limF;   // pool sizes in use (<= MEM's; smaller only in tests of the repeat path)
  unsigned rrank;
  unsigned rc[3];
  float rp[3];
  V3 normal;
  float depth;
  unsigned nextsv;
  unsigned long long hullLo, hullHi, freeLo, freeHi;   // faces 0..63 / 64..127 on the hull, in the stock
  unsigned hullCount;
  unsigned stamp;
  EpaProf prof;
  // horizon
  unsigned hnf;     // horizon edges met by this iteration's expansions (m->hz)
  unsigned avail;   // free face slots as the reference's expand would find them at this point of the traversal: every face
                    // it has removed so far is free again, every horizon edge met so far has taken one

  TOHC_DEV V3 svW(unsigned sv) const { return ld3(m->svw[sv]); }
  static TOHC_DEV unsigned byteOf(uint32_t w, unsigned k) { return (w >> (8 * k)) & 255u; }
  TOHC_DEV void bind(unsigned fa, unsigned ea, unsigned fb, unsigned eb) {
    ((uint8_t *)&m->fe[fa])[ea] = (uint8_t)eb; ((uint8_t *)&m->ff[fa])[ea] = (uint8_t)fb;
    ((uint8_t *)&m->fe[fb])[eb] = (uint8_t)ea; ((uint8_t *)&m->ff[fb])[eb] = (uint8_t)fa;
  }
  TOHC_DEV bool onHull(unsigned f) const { return (((f < 64 ? hullLo : hullHi) >> (f & 63)) & 1ull) != 0; }
  TOHC_DEV void hullAppend(unsigned f) {   // append(m_hull, face)
    if (f < 64) hullLo |= 1ull << f; else hullHi |= 1ull << (f - 64);
    ++hullCount;
    m->fstamp[f] = (uint16_t)(++stamp);
  }
  TOHC_DEV void hullToStock(unsigned f) {  // remove(m_hull, face); append(m_stock, face)
    if (f < 64) { hullLo &= ~(1ull << f); freeLo |= 1ull << f; } else { hullHi &= ~(1ull << (f - 64)); freeHi |= 1ull << (f - 64); }
    --hullCount;
  }
  TOHC_DEV void init(unsigned lv = MEM::kNV, unsigned lf = MEM::kNF) {  // [epa:583-593]
    limV = lv < (unsigned)MEM::kNV ? lv : (unsigned)MEM::kNV;
    limF = lf < (unsigned)MEM::kNF ? lf : (unsigned)MEM::kNF;
    status = E_FAILED;
    normal = mk(0, 0, 0);
    depth = 0;
    nextsv = 0;
    stamp = 0;
    hullLo = hullHi = 0;
    hullCount = 0;
    freeLo = limF >= 64 ? ~0ull : (1ull << (limF & 63)) - 1ull;
    freeHi = limF >= 128 ? ~0ull : (limF > 64 ? (1ull << (limF & 63)) - 1ull : 0ull);
    exhausted = false;
  }
  TOHC_DEV bool getedgedist(V3 fnrm, V3 a, V3 b, float &dist) {  // [epa:698-733]
    V3 ba = b - a;
    V3 n_ab = cross(ba, fnrm);
    float a_dot_nab = dot(a, n_ab);
    if (a_dot_nab < 0) {
      float ba_l2 = len2(ba);
      float a_dot_ba = dot(a, ba);
      float b_dot_ba = dot(b, ba);
      if (a_dot_ba > 0) dist = len(a);
      else if (b_dot_ba < 0) dist = len(b);
      else {
        float a_dot_b = dot(a, b);
        float t = (len2(a) * len2(b) - a_dot_b * a_dot_b) / ba_l2;
        dist = sqrtf(t > 0.0f ? t : 0.0f);
      }
      return true;
    }
    return false;
  }
  // [epa:734-778].  The reference takes the stock list's root; WHICH free slot a face gets is invisible in the results
  // (faces are told apart by geometry, adjacency and arrival stamp only), so the lowest free slot is taken.
  TOHC_DEV unsigned newface(unsigned a, unsigned b, unsigned c, bool forced) {
    if ((freeLo | freeHi) != 0) {
      unsigned face;
      if (freeLo) { face = (unsigned)__builtin_ctzll(freeLo); freeLo &= freeLo - 1; }
      else { face = 64u + (unsigned)__builtin_ctzll(freeHi); freeHi &= freeHi - 1; }
      hullAppend(face);
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
      if (face > prof.maxface) prof.maxface = face;
#endif
      m->fcp[face] = a | (b << 8) | (c << 16);   // pass = 0
      V3 aw = svW(a), bw = svW(b), cw = svW(c);
      V3 n = cross(bw - aw, cw - aw);
      float l = len(n);
      bool v = l > 0.0001f;
      if (v) {
        float d;
        if (!(getedgedist(n, aw, bw, d) || getedgedist(n, bw, cw, d) || getedgedist(n, cw, aw, d))) d = dot(aw, n) / l;
        n = vdiv(n, l);
        float *o = m->fnd[face];
        o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = d;
        if (forced || (d >= -0.00001f)) return face;
        else status = E_NONCONVEX;
      } else {
        status = E_DEGENERATED;
      }
      hullToStock(face);
      return TOHC_NIL;
    }
    status = E_OUTOFFACES;
    if (limF < TOHC_EPA_MAX_FACES) exhausted = true;
    return TOHC_NIL;
  }
  // findbest [epa:779-793] walks the hull list from its root (newest face first) and keeps the first strict
  // minimum of d*d.  Same result without the pointer chase: every lane looks at its share of the face array, and
  // among equal minima the newest face (largest stamp) wins.  (A NaN distance never wins; the reference would return
  // the list root if the ROOT's were NaN -- finite shapes cannot produce one.)
  TOHC_DEV unsigned findbest() {
    float best = INFINITY;
    int bkey = -1;   // (stamp << 8) | face; larger = earlier in the list
    for (unsigned f = (unsigned)Grp<G>::lane(); f < (unsigned)MEM::kNF; f += G) {
      if (!onHull(f)) continue;
      const float d = m->fnd[f][3];
      const float sqd = d * d;
      const int key = ((int)m->fstamp[f] << 8) | (int)f;
      if (sqd < best || (sqd == best && key > bkey)) { best = sqd; bkey = key; }
    }
    // group-wide arg-min with the same tie rule: arg-max of (-d*d, key); a lane that saw nothing holds (-inf, -1)
    float nb = -best;
    Grp<G>::argmax_hi(nb, bkey);
    if (bkey < 0) return hullLo ? (unsigned)__builtin_ctzll(hullLo) : 64u + (unsigned)__builtin_ctzll(hullHi);
    return (unsigned)(bkey & 255);
  }
  // expand [epa:794-827] in two phases.  The reference's recursion makes each new face the moment it meets a horizon
  // edge; nothing it does afterwards in the same iteration reads a new face (the traversal walks visible faces and
  // tests the faces beyond the horizon, new faces hang off the far side of horizon edges), so here
  //   traverse():    the same walk on an explicit stack (stage 0: entry, 1: first child returned, 2: second child
  //                  returned) only NOTES the horizon edges, in visiting order, and counts the free slots exactly as
  //                  the reference would find them (avail) -- running out of faces is detected at the same edge;
  //   makeHorizon(): makes all the new faces at once, one lane per face, and binds the fan.
  // A failure the reference meets half way (degenerate / non-convex new face, out of faces, a face reached twice) ends
  // its iteration with what it had before (E_INVALIDHULL: the result is the previous best face), and it does here.
  // The two entries at the top of the stack live in registers: `top` is the entry being worked on, `below` its parent.
  // A push stores the parent (LDS write, nobody waits) and a pop takes `below` and re-reads the entry under it -- a read
  // whose result is not needed before the NEXT pop, so it travels with the face reads instead of being a round trip of
  // its own at the head of every step of the walk (it was: 5 dependent LDS round trips per visible face, 2 per leaf;
  // now 2 and 1).
  TOHC_DEV bool traverse(unsigned pass, unsigned w, unsigned f0, unsigned e0) {
    int sp = 1;
    bool ret = false;
    unsigned top = f0 | (e0 << 8), below = 0;
    const V3 ww = svW(w);
#define TOHC_EPA_POP() do { --sp; top = below; below = sp >= 2 ? (unsigned)m->stk[sp - 2] : 0u; } while (0)
#define TOHC_EPA_PUSH(parent, child) do { m->stk[sp - 1] = (uint16_t)(parent); below = (parent); top = (child); ++sp; } while (0)
    while (sp > 0) {
      const unsigned ent = top;
      const unsigned f = ent & 255u, e = (ent >> 8) & 3u, stage = ent >> 10;
      const unsigned e1 = (e + 1) % 3, e2 = (e + 2) % 3;
      if (stage == 0) {
        // everything this visit can need, issued together: one LDS round trip
        const uint32_t cp = m->fcp[f], fw = m->ff[f], ew = m->fe[f];
        const float *nd = m->fnd[f];
        const float nx = nd[0], ny = nd[1], nz = nd[2], fd = nd[3];
        if ((cp >> 24) != pass) {
          if ((dot(mk(nx, ny, nz), ww) - fd) < -0.00001f) {
            if (avail != 0) {   // newface [epa:801] would find a slot
              --avail;
              m->hz[hnf++] = (uint16_t)(f | (e << 7));   // (hnf <= the pool's faces: each entry has taken a free slot)
              ret = true;
            } else {
              status = E_OUTOFFACES;
              if (limF < TOHC_EPA_MAX_FACES) exhausted = true;
              ret = false;
            }
            TOHC_EPA_POP();
          } else {
            ((uint8_t *)&m->fcp[f])[3] = (uint8_t)pass;
            if (sp >= MEM::kNF) { ret = false; TOHC_EPA_POP(); continue; }  // cannot happen: depth <= hull faces
            TOHC_EPA_PUSH(f | (e << 8) | (1u << 10), byteOf(fw, e1) | (byteOf(ew, e1) << 8));
          }
        } else {
          ret = false;
          TOHC_EPA_POP();
        }
      } else if (stage == 1) {
        if (ret) {
          const uint32_t fw = m->ff[f], ew = m->fe[f];
          TOHC_EPA_PUSH(f | (e << 8) | (2u << 10), byteOf(fw, e2) | (byteOf(ew, e2) << 8));
        } else {
          ret = false;
          TOHC_EPA_POP();
        }
      } else {
        if (ret) {
          hullToStock(f);
          ++avail;
          ret = true;
        } else
          ret = false;
        TOHC_EPA_POP();
      }
    }
#undef TOHC_EPA_POP
#undef TOHC_EPA_PUSH
    return ret;
  }
  // The three expansions of one iteration without the recursion's float work on the walk.  Whether w sees a face,
  // (dot(n, w) - d) < -eps or not, does not depend on when the walk reaches it: ONE LANE PER FACE settles that for the
  // whole hull first (a 128-bit mask).  The walk itself is then integer work: a face w sees is entered once (a second
  // arrival is the reference's `pass` test failing: false), its second neighbour waits on the stack while the first is
  // followed; a face w does not see is a horizon edge, noted in visiting order, and the walk goes on with the entry on
  // top of the stack -- whose value sits in a register (the stack in LDS holds every entry; a pop re-reads the one
  // below off the critical path).  Per face seen: one LDS round trip (its neighbours); per horizon edge: none.
  // What the walk does not reproduce is WHEN the recursion returns a seen face to the stock, i.e. how many free slots
  // newface finds at a given horizon edge; so it only answers when the free slots at the start cover the whole horizon
  // (no edge can run out then), and says EX_SLOW otherwise: the caller runs traverse(), the exact walk, instead.
  // EX_OK: m->hz[0 .. hnf) = the horizon; (rmLo, rmHi) = the faces the recursion has returned to the stock.
  enum { EX_OK, EX_FAIL, EX_SLOW };
  TOHC_DEV int expandFast(unsigned w, unsigned best, uint32_t bfw, uint32_t bew, unsigned long long &rmLo, unsigned long long &rmHi) {
    const V3 ww = svW(w);
    unsigned long long vLo = 0, vHi = 0;
    for (unsigned f = (unsigned)Grp<G>::lane(); f < (unsigned)MEM::kNF; f += G) {
      if (!onHull(f)) continue;
      const float *nd = m->fnd[f];
      const bool beyond = (dot(mk(nd[0], nd[1], nd[2]), ww) - nd[3]) < -0.00001f;   // [epa:800]
      if (!beyond) { if (MEM::kNF <= 64 || f < 64) vLo |= 1ull << (f & 63u); else vHi |= 1ull << (f - 64); }
    }
    vLo = Grp<G>::orAll64(vLo);
    if (MEM::kNF > 64) vHi = Grp<G>::orAll64(vHi);
    unsigned long long sLo = best < 64 ? 1ull << best : 0ull, sHi = best < 64 ? 0ull : 1ull << (best - 64);   // entered (best carries `pass` already)
    const unsigned slots = avail;
    unsigned sp = 0, top = 0;
    for (unsigned j = 0; j < 3; ++j) {
      unsigned cur = byteOf(bfw, j) | (byteOf(bew, j) << 8);
      for (;;) {
        const unsigned f = cur & 255u, e = cur >> 8;
        const unsigned long long bit = 1ull << (f & 63u);
        const bool hi = MEM::kNF > 64 && f >= 64;   // (a pool of <= 64 faces: the upper words are dead code)
        if (((hi ? sHi : sLo) & bit) != 0) return EX_FAIL;            // f->pass == pass [epa:797]
        if (((hi ? vHi : vLo) & bit) != 0) {
          if (hi) sHi |= bit; else sLo |= bit;
          const uint32_t fw = m->ff[f], ew = m->fe[f];
          const unsigned e1 = (e + 1) % 3, e2 = (e + 2) % 3;
          top = byteOf(fw, e2) | (byteOf(ew, e2) << 8);
          m->stk[sp++] = (uint16_t)top;                              // (sp < faces entered <= the pool's faces)
          cur = byteOf(fw, e1) | (byteOf(ew, e1) << 8);
        } else {
          if (hnf >= slots) return EX_SLOW;                            // the exact walk decides whether the stock runs out here
          m->hz[hnf++] = (uint16_t)(f | (e << 7));
          if (sp == 0) break;
          cur = top;
          --sp;
          top = sp > 0 ? (unsigned)m->stk[sp - 1] : 0u;
        }
      }
    }
    const unsigned long long bb = 1ull << (best & 63u);
    rmLo = best < 64 ? sLo & ~bb : sLo;
    rmHi = best < 64 ? sHi : sHi & ~bb;
    return EX_OK;
  }
  // The fan of new faces over the horizon m->hz[0 .. hnf), w = the new vertex: face k = (c[e+1], c[e], w) of horizon
  // entry k [epa:801], bound to the face beyond its edge [epa:804], to its neighbours in the fan [epa:805] and round the
  // end [epa:672]; arrival stamps in horizon order, as the reference's appends.  One lane per new face.
  TOHC_DEV bool makeHorizon(unsigned w) {
    const unsigned n = hnf, stamp0 = stamp;
    // slots: the lowest free ones (which slot a face gets is invisible in the results)
    for (unsigned k = 0; k < n; ++k) {
      unsigned face;
      if (freeLo) { face = (unsigned)__builtin_ctzll(freeLo); freeLo &= freeLo - 1; hullLo |= 1ull << face; }
      else { face = (unsigned)__builtin_ctzll(freeHi); freeHi &= freeHi - 1; hullHi |= 1ull << face; face += 64u; }
      m->hz[k] = (uint16_t)((m->hz[k] & 511u) | (face << 9));   // (every lane of the group writes the same value)
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
      if (face > prof.maxface) prof.maxface = face;
#endif
    }
    hullCount += n;
    stamp += n;
    Grp<G>::fence();
    const V3 cw = svW(w);
    bool bad = false;
    int st = status;
    for (unsigned k = (unsigned)Grp<G>::lane(); k < n; k += G) {
      const unsigned hk = m->hz[k], hn = m->hz[k + 1 < n ? k + 1 : 0], hp = m->hz[k ? k - 1 : n - 1];
      const unsigned f = hk & 127u, e = (hk >> 7) & 3u, face = hk >> 9;
      const uint32_t cp = m->fcp[f];
      const unsigned a = byteOf(cp, (e + 1) % 3), b = byteOf(cp, e);
      m->fcp[face] = a | (b << 8) | (w << 16);   // pass = 0
      m->fstamp[face] = (uint16_t)(stamp0 + k + 1);
      m->ff[face] = f | ((hn >> 9) << 8) | ((hp >> 9) << 16);   // bind(nf, 0, f, e); bind(nf, 1, next, 2); bind(prev, 1, nf, 2)
      m->fe[face] = e | (2u << 8) | (1u << 16);
      ((uint8_t *)&m->ff[f])[e] = (uint8_t)face;
      ((uint8_t *)&m->fe[f])[e] = 0;
      // newface [epa:734-778], not forced
      const V3 aw = svW(a), bw = svW(b);
      V3 nrm = cross(bw - aw, cw - aw);
      const float l = len(nrm);
      if (l > 0.0001f) {
        float d;
        if (!(getedgedist(nrm, aw, bw, d) || getedgedist(nrm, bw, cw, d) || getedgedist(nrm, cw, aw, d))) d = dot(aw, nrm) / l;
        nrm = vdiv(nrm, l);
        float *o = m->fnd[face];
        o[0] = nrm.x; o[1] = nrm.y; o[2] = nrm.z; o[3] = d;
        if (!(d >= -0.00001f)) { bad = true; st = E_NONCONVEX; }
      } else {
        bad = true; st = E_DEGENERATED;
      }
    }
    Grp<G>::fence();
    if (Grp<G>::any(bad)) { status = bad ? st : E_NONCONVEX; return false; }   // (the caller ends the run: E_INVALIDHULL)
    return true;
  }

  TOHC_DEV int evaluate(Gjk2<G, MEM> &g, V3 guess) {  // [epa:594-697]
    const unsigned s = g.current;
    TOHC_PROF_T0(tenc);
    const bool enclosed = (g.rk(s) > 1) && g.encloseOrigin();
    TOHC_PROF_ACC(prof, 4, tenc);
    if (enclosed) {
      // (the reference first returns a previous run's hull to the stock; here every evaluation starts from init())
      status = E_VALID;
      nextsv = 0;
      unsigned c0 = m->sc[s][0], c1 = m->sc[s][1], c2 = m->sc[s][2], c3 = m->sc[s][3];
      if (det3(svW(c0) - svW(c3), svW(c1) - svW(c3), svW(c2) - svW(c3)) < 0) {
        unsigned t = c0; c0 = c1; c1 = t;
        m->sc[s][0] = (uint8_t)c0; m->sc[s][1] = (uint8_t)c1;
        float tp = m->sp[s][0]; m->sp[s][0] = m->sp[s][1]; m->sp[s][1] = tp;
      }
      TOHC_PROF_T0(tnf);
      unsigned t0 = newface(c0, c1, c2, true);
      unsigned t1 = newface(c1, c0, c3, true);
      unsigned t2 = newface(c2, c1, c3, true);
      unsigned t3 = newface(c0, c2, c3, true);
      TOHC_PROF_ACC(prof, 5, tnf);
      if (hullCount == 4) {
        unsigned best = findbest();
        TOHC_STAT(emuStats().last_d0 = (double)m->fnd[best][3];)
        V3 outerN = ld3(m->fnd[best]);
        float outerD = m->fnd[best][3];
        uint32_t ocp = m->fcp[best];
        unsigned pass = 0, iterations = 0;
        bind(t0, 0, t1, 0);
        bind(t0, 1, t2, 0);
        bind(t0, 2, t3, 0);
        bind(t1, 1, t3, 2);
        bind(t1, 2, t2, 1);
        bind(t2, 2, t3, 1);
        status = E_VALID;
        for (; iterations < 255; ++iterations) {
          if (nextsv < limV) {
            hnf = 0;
            avail = (unsigned)(__builtin_popcountll(freeLo) + __builtin_popcountll(freeHi));
            unsigned w = 4 + nextsv++;
            bool valid = true;
            ((uint8_t *)&m->fcp[best])[3] = (uint8_t)(++pass);
            const uint32_t bfw = m->ff[best], bew = m->fe[best];
            V3 bn = ld3(m->fnd[best]);
            const float bd = m->fnd[best][3];
            TOHC_PROF_T0(ts);
            g.support(bn, w);
            TOHC_PROF_ACC(prof, 1, ts);
            float wdist = dot(bn, svW(w)) - bd;
            if (wdist > 0.0001f) {
              TOHC_PROF_T0(te);
              // best's own adjacency is not rewritten while its neighbours are expanded (binds touch the new faces
              // and the faces beyond the horizon), so the words read above stay current
              unsigned long long rmLo = 0, rmHi = 0;
              const int ex = TOHC_EPA_FAST_EXPAND ? expandFast(w, best, bfw, bew, rmLo, rmHi) : EX_SLOW;
              if (ex == EX_OK) {   // the faces the recursion has removed, back to the stock
                hullLo &= ~rmLo; hullHi &= ~rmHi; freeLo |= rmLo; freeHi |= rmHi;
                hullCount -= (unsigned)(__builtin_popcountll(rmLo) + __builtin_popcountll(rmHi));
              } else if (ex == EX_FAIL) {
                valid = false;
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
                ++prof.nfail;
#endif
              } else {
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
                ++prof.nslow;
#endif
                hnf = 0;
                for (unsigned j = 0; (j < 3) && valid; ++j) valid &= traverse(pass, w, byteOf(bfw, j), byteOf(bew, j));
              }
              TOHC_PROF_T0(tm);
              if (valid && (hnf >= 3)) valid = makeHorizon(w);
              TOHC_PROF_ACC(prof, 7, tm);
              TOHC_PROF_ACC(prof, 2, te);
              if (valid && (hnf >= 3)) {
                hullToStock(best);
                TOHC_PROF_T0(tf);
                best = findbest();
                TOHC_PROF_ACC(prof, 3, tf);
                outerN = ld3(m->fnd[best]);
                outerD = m->fnd[best][3];
                ocp = m->fcp[best];
              } else { status = E_INVALIDHULL; break; }
            } else { status = E_ACCURACY; break; }
          } else { status = E_OUTOFVERTICES; if (limV < TOHC_EPA_MAX_VERTICES) exhausted = true; break; }
        }
        TOHC_STAT(emuStats().epa_calls++; emuStats().epa_iters += iterations; emuStats().epa_faces += hullCount;
                  emuStats().hist_epa[iterations < 71 ? iterations : 71]++; emuStats().status[status]++;)
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
        prof.nsv = nextsv;
#endif
        const unsigned oc0 = byteOf(ocp, 0), oc1 = byteOf(ocp, 1), oc2 = byteOf(ocp, 2);
        V3 projection = outerN * outerD;
        normal = outerN;
        depth = outerD;
        rrank = 3;
        rc[0] = oc0; rc[1] = oc1; rc[2] = oc2;
        rp[0] = len(cross(svW(oc1) - projection, svW(oc2) - projection));
        rp[1] = len(cross(svW(oc2) - projection, svW(oc0) - projection));
        rp[2] = len(cross(svW(oc0) - projection, svW(oc1) - projection));
        float sum = rp[0] + rp[1] + rp[2];
        rp[0] /= sum; rp[1] /= sum; rp[2] /= sum;
        return status;
      }
    }
    status = E_FALLBACK;
    normal = -guess;
    float nl = len(normal);
    if (nl > 0) normal = vdiv(normal, nl);
    else normal = mk(1, 0, 0);
    depth = 0;
    rrank = 1;
    rc[0] = m->sc[s][0];
    rp[0] = 1;
    return status;
  }
};

template <int G>
TOHC_DEV void minkInit(Mink<G> &m, const Cvx<G> *s0, const Xf &w0, const Cvx<G> *s1, const Xf &w1, bool margins) {
  m.s0 = s0; m.s1 = s1;
  m.toshape1 = transposeTimes(w1.b, w0.b);
  m.toshape0 = xfInverseTimes(w0, w1);
  m.margins = margins;
}

// btGjkEpaPenetrationDepthSolver::calcPenDepth [pds:24-64] = Penetration [epa:903-945], else Distance [epa:865-900],
// cut into two stages so that the device can run them in different kernels:
//   stage 1 (penStage1): the GJK2 of Penetration (with margins).  Origin not inside -> Distance's GJK2 (without) and
//            the result is final.  Origin inside -> EPA is needed; the GJK2 state goes into a PenHand.
//   stage 2 (penStage2): EPA from that state, witness points.
struct PenHand {
  G2Mem mem;
  unsigned rank0, rank1, nfree, current;
};
template <class MA, class MB>
TOHC_DEV void g2Copy(MA *dst, const MB *src) {
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < 3; ++k) { dst->svd[i][k] = src->svd[i][k]; dst->svw[i][k] = src->svw[i][k]; }
    dst->freeL[i] = src->freeL[i];
    for (int q = 0; q < 2; ++q) { dst->sc[q][i] = src->sc[q][i]; dst->sp[q][i] = src->sp[q][i]; }
  }
}
// Stage 1 in its two halves (the device runs them in different kernels: the second one is needed by one pair in nine,
// and a lane-per-pair wavefront that holds a single such pair sits through all of it):
//   penInside():   the GJK2 of Penetration (with margins); true = the origin is inside, EPA is needed, g holds the state
//   penDistance(): Distance's GJK2 (without margins) and its witness points; isValid2 / v / wA / wB are final
template <int G, class MEM>
TOHC_DEV bool penInside(MEM *mem, const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, Gjk2<G, MEM> &g) {
  V3 guess = tA.o - tB.o;
  Mink<G> shape;
  minkInit<G>(shape, &A, tA, &B, tB, true);
  g.m = mem;
  return g.evaluate(shape, -guess) == G2_INSIDE;
}
template <int G, class MEM>
TOHC_DEV void penDistance(MEM *mem, const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, Gjk2<G, MEM> &g, bool &isValid2, V3 &v, V3 &wA, V3 &wB) {
  V3 guess = tA.o - tB.o;
  Mink<G> shape;
  isValid2 = false;
  minkInit<G>(shape, &A, tA, &B, tB, false);
  g.m = mem;
  int st = g.evaluate(shape, guess);
  if (st == G2_VALID) {
    V3 a = mk(0, 0, 0), b = mk(0, 0, 0);
    const unsigned s = g.current;
    for (unsigned i = 0; i < g.rk(s); ++i) {
      float p = mem->sp[s][i];
      V3 d = ld3(mem->svd[mem->sc[s][i]]);
      a = a + minkSupport0<G>(shape, d) * p;
      b = b + minkSupport1<G>(shape, -d) * p;
    }
    wA = xfApply(tA, a);
    wB = xfApply(tA, b);
    V3 nrm = a - b;
    float dist = len(nrm);
    v = vdiv(nrm, dist > 0.0001f ? dist : 1);
  }
}
// both halves: returns true when EPA is needed (g holds the state); else isValid2 / v / wA / wB are final
template <int G, class MEM>
TOHC_DEV bool penStage1(MEM *mem, const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, Gjk2<G, MEM> &g, bool &isValid2, V3 &v, V3 &wA, V3 &wB) {
  isValid2 = false;
  if (penInside<G, MEM>(mem, A, B, tA, tB, g)) return true;
  penDistance<G, MEM>(mem, A, B, tA, tB, g, isValid2, v, wA, wB);
  return false;
}
// g: the state penStage1 left (restored into an EpaMem when the stages run in different kernels)
// returns false when MEM is smaller than the reference's pools and ran out (nothing is written then)
template <int G, class MEM>
TOHC_DEV bool penStage2(MEM *mem, const Xf &tA, const Xf &tB, Gjk2<G, MEM> &g, bool &isValid2, V3 &v, V3 &wA, V3 &wB, EpaProf *prof = nullptr,
                        unsigned limV = MEM::kNV, unsigned limF = MEM::kNF) {
  V3 guess = tA.o - tB.o;
  Epa<G, MEM> e;
  e.m = mem;
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
  for (int k = 0; k < 8; ++k) { e.prof.t[k] = 0; e.prof.n[k] = 0; }
  e.prof.maxface = 0; e.prof.nsv = 0; e.prof.nslow = 0; e.prof.nfail = 0;
#endif
  TOHC_PROF_T0(ti);
  e.init(limV, limF);
  TOHC_PROF_ACC(e.prof, 0, ti);
  TOHC_PROF_T0(tev);
  e.evaluate(g, -guess);  // never leaves E_FAILED: every path sets a status, so the reference's Distance fall-back is dead here
  TOHC_PROF_ACC(e.prof, 6, tev);
#if defined(TOHC_EPA_PROF) && defined(__HIPCC__)
  if (prof) *prof = e.prof;
#endif
  if (e.exhausted) return false;
  V3 w0 = mk(0, 0, 0);
  for (unsigned i = 0; i < e.rrank; ++i) {
    unsigned ci = e.rc[0];
    float pi = e.rp[0];
    if (i == 1) { ci = e.rc[1]; pi = e.rp[1]; }
    if (i == 2) { ci = e.rc[2]; pi = e.rp[2]; }
    w0 = w0 + minkSupport0<G>(g.shape, ld3(mem->svd[ci])) * pi;
  }
  wA = xfApply(tA, w0);
  wB = xfApply(tA, w0 - e.normal * e.depth);
  v = -e.normal;
  isValid2 = true;
  return true;
}

// second half of getClosestPointsNonVirtual [gjk:334-449]: penetration / degenerate handling and the output filter
struct PairOut {
  bool valid;
  V3 normalOnB, pointOnB;
  float distance;
};

// (marginA + marginB + gContactBreakingThreshold)^2: the pair detector's m_maximumDistanceSquared [cca:370-373]
template <int G>
TOHC_DEV float pairMaxDist2(const Cvx<G> &A, const Cvx<G> &B, float breaking) {
  float md = cvxMargin<G>(A) + cvxMargin<G>(B) + breaking;
  return md * md;
}
// Two PLAIN spheres never reach GJK: the dispatch matrix holds btSphereSphereCollisionAlgorithm for that pair of shape
// types (btDefaultCollisionConfiguration.cpp:219-222; a compound's child is dispatched by the child's type,
// btCompoundCollisionAlgorithm.cpp:65).  A swept sphere is a CastHullShape, a custom convex type: that one goes to GJK.
template <int G>
TOHC_DEV bool isSphereSphere(const Cvx<G> &A, const Cvx<G> &B) { return A.kind == KIND_SPHERE && B.kind == KIND_SPHERE && !A.cast && !B.cast; }
// btSphereSphereCollisionAlgorithm::processCollision (btSphereSphereCollisionAlgorithm.cpp:42-89): no contact at all
// while the surfaces are apart
TOHC_DEV void sphereSphere(V3 oA, float radius0, V3 oB, float radius1, PairOut &o) {
  o.valid = false;
  const V3 diff = oA - oB;
  const float l = len(diff);
  if (l > (radius0 + radius1)) return;
  o.distance = l - (radius0 + radius1);
  o.normalOnB = mk(1, 0, 0);
  if (l > FLT_EPSILON) o.normalOnB = vdiv(diff, l);
  o.pointOnB = oB + o.normalOnB * radius1;
  o.valid = true;
}
// The one pair kind whose reference cull is not grown by d on either side: a swept sphere against the sphere child of
// a COMPOUND.  The compound's child test [btCompoundCollisionAlgorithm.cpp:128-132] takes btSphereShape::getAabb
// (centre +- radius; the SHAPE_EXPANSION patch sits in btTransformAabb, which spheres do not call) for the child and
// CastHullShape::getAabb [bcc:850-856] = the union of the same at both ends of the sweep for the swept sphere; pairs it
// rejects never reach GJK, though GJK would report a contact at a distance in (0, d].  true = the reference culls.
TOHC_DEV bool castSphereChildCulled(V3 o0, V3 o1, float rq, V3 ox, float rx) {
  const V3 q = mk(rq, rq, rq), x = mk(rx, rx, rx);
  Aabb bs = aabbUnion(Aabb{o0 - q, o0 + q}, Aabb{o1 - q, o1 + q});
  Aabb bc{ox - x, ox + x};
  return !aabbOverlap(bs, bc);
}

// [gjk:349-416] what getClosestPointsNonVirtual does with calcPenDepth's answer
TOHC_DEV void gjkFinishTail(bool isValid2, V3 sep, V3 tmpA, V3 tmpB, GjkState &st) {
  if (isValid2) {
    V3 tmpN = tmpB - tmpA;
    float lenSqr = len2(tmpN);
    if (lenSqr <= (FLT_EPSILON * FLT_EPSILON)) { tmpN = sep; lenSqr = len2(sep); }
    if (lenSqr > (FLT_EPSILON * FLT_EPSILON)) {
      tmpN = vdiv(tmpN, sqrtf(lenSqr));
      float distance2 = -len(tmpA - tmpB);
      if (!st.isValid || (distance2 < st.distance)) {
        st.distance = distance2; st.pointOnB = tmpB; st.normalInB = tmpN; st.isValid = true;
      }
    }
  } else {
    if (len2(sep) > 0.0f) {
      const float margin = st.mA + st.mB;
      float distance2 = margin != 0.0f ? len(tmpA - tmpB) - margin : len(tmpA - tmpB);
      if (!st.isValid || (distance2 < st.distance)) {
        st.distance = distance2;
        st.pointOnB = st.mB != 0.0f ? tmpB + sep * st.mB : tmpB;   // [gjk:404-405]
        st.normalInB = vdiv(sep, len(sep));
        st.isValid = true;
      }
    }
  }
}
// the operand transforms calcPenDepth sees: shifted by the midpoint offset [gjk:112-115]
TOHC_DEV void penLocal(const Xf &tA, const Xf &tB, V3 offset, Xf &lA, Xf &lB) {
  lA = tA; lB = tB;
  lA.o = lA.o - offset;
  lB.o = lB.o - offset;
}

// hand-over between the stages when they run in different kernels
template <int G, class MEM>
TOHC_DEV void penSave(PenHand &h, const Gjk2<G, MEM> &g) {
  g2Copy(&h.mem, g.m);
  h.rank0 = g.rank0; h.rank1 = g.rank1; h.nfree = g.nfree; h.current = g.current;
}
template <int G, class MEM>
TOHC_DEV void penRestore(MEM *mem, const PenHand &h, const Cvx<G> &A, const Cvx<G> &B, const Xf &lA, const Xf &lB, Gjk2<G, MEM> &g) {
  g2Copy(mem, &h.mem);
  g.m = mem;
  g.rank0 = h.rank0; g.rank1 = h.rank1; g.nfree = h.nfree; g.current = h.current;
  g.status = G2_INSIDE;
  g.distance = 0;
  g.ray = mk(0, 0, 0);
  minkInit<G>(g.shape, &A, lA, &B, lB, true);
}

// both stages back to back on one EpaMem (host emulation; the device runs k_pen then k_epa)
template <int G>
TOHC_DEV void gjkFinishEpa(EpaMem *mem, const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, GjkState &st) {
  Xf lA, lB;
  penLocal(tA, tB, st.offset, lA, lB);
  V3 tmpA = mk(0, 0, 0), tmpB = mk(0, 0, 0), sep = mk(0, 0, 0);
  bool isValid2 = false;
  Gjk2<G, EpaMem> g;
  if (penStage1<G, EpaMem>(mem, A, B, lA, lB, g, isValid2, sep, tmpA, tmpB)) penStage2<G>(mem, lA, lB, g, isValid2, sep, tmpA, tmpB);
  gjkFinishTail(isValid2, sep, tmpA, tmpB, st);
}

TOHC_DEV void gjkOutput(const GjkState &st, float maxDist2, PairOut &o) {
  o.valid = false;
  if (st.isValid && ((st.distance < 0) || (st.distance * st.distance < maxDist2))) {
    o.valid = true;
    o.normalOnB = st.normalInB;
    o.pointOnB = st.pointOnB + st.offset;
    o.distance = st.distance;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GetAverageSupport [bcc:98-131] for a box (8 vertices, btBoxShape::getVertex) or a hull.  The reference sums the
// vertices serially in index order with a running maximum, which is order dependent; the group replays exactly that
// scan: vertex i is broadcast from its owning lane in index order (only for kept contacts, so the cost is small).
// ---------------------------------------------------------------------------------------------------------------
struct AvgAcc { V3 sum; float cnt, maxS; };
TOHC_DEV void avgStep(AvgAcc &a, V3 pt, V3 nl) {
  const float EPS = 1e-3f;
  float sup = dot(pt, nl);
  if (sup > a.maxS + EPS) { a.cnt = 1; a.sum = pt; a.maxS = sup; }
  else if (sup < a.maxS - EPS) {}
  else { a.cnt += 1; a.sum = a.sum + pt; }
}
// hull vertices in index order; the register slots are indexed by compile-time constants only (a dynamic slot index
// sends the whole vertex array through scratch memory: 80 bytes per LANE and hit, 650 MB per launch of k_finish)
#if defined(__HIPCC__)
template <int G, int SRC>
struct AvgSrc {   // vertices base + SRC .. base + G - 1 of one register slot, source lanes as compile-time constants
  static TOHC_DEV void run(float x, float y, float z, int base, int n, V3 nl, AvgAcc &a) {
    if (base + SRC < n) avgStep(a, mk(Grp<G>::template bcastC<SRC>(x), Grp<G>::template bcastC<SRC>(y), Grp<G>::template bcastC<SRC>(z)), nl);
    if constexpr (SRC + 1 < G) AvgSrc<G, SRC + 1>::run(x, y, z, base, n, nl, a);
  }
};
#endif
template <int G>
TOHC_DEV void avgHull(const Cvx<G> &c, V3 nl, AvgAcc &a) {
#pragma unroll
  for (int s = 0; s < Cvx<G>::VPL; ++s) {
    if (s * G >= c.n) break;
#if defined(__HIPCC__)
    if constexpr (G == 4) {
      AvgSrc<G, 0>::run(c.x[s], c.y[s], c.z[s], s * G, c.n, nl, a);
      continue;
    }
#endif
    const int lim = c.n - s * G < G ? c.n - s * G : G;
    for (int src = 0; src < lim; ++src) avgStep(a, mk(Grp<G>::bcast(c.x[s], src), Grp<G>::bcast(c.y[s], src), Grp<G>::bcast(c.z[s], src)), nl);
  }
}
template <>
TOHC_DEV void avgHull<1>(const Cvx<1> &c, V3 nl, AvgAcc &a) {
  for (int i = 0; i < c.n; ++i) avgStep(a, cvxVertex<1>(c, i), nl);
}
template <int G>
TOHC_DEV void averageSupport(const Cvx<G> &c, V3 nl, float &outsup, V3 &outpt) {
  if (c.kind == KIND_CYLZ || c.kind == KIND_SPHERE) {  // not polyhedral: localGetSupportingVertexWithoutMargin itself [bcc:124-129]
    outpt = c.kind == KIND_SPHERE ? mk(0, 0, 0) : cylZSupport(c.dims, nl);
    outsup = dot(nl, outpt);
    return;
  }
  AvgAcc a;
  a.sum = mk(0, 0, 0); a.cnt = 0; a.maxS = -1000;
  if (c.kind == KIND_BOX) {
    const V3 he = c.dims;
    for (int i = 0; i < 8; ++i)
      avgStep(a, mk(he.x * (1 - (i & 1)) - he.x * (i & 1), he.y * (1 - ((i & 2) >> 1)) - he.y * ((i & 2) >> 1),
                    he.z * (1 - ((i & 4) >> 2)) - he.z * ((i & 4) >> 2)), nl);
  } else {
    avgHull<G>(c, nl, a);
  }
  outsup = a.maxS;
  outpt = vdiv(a.sum, a.cnt);
}

// CastCollisionCollector::addSingleResult [bcc:894-972].  In: the contact as Bullet reported it (A/B = GJK roles,
// castFirst <=> the swept shape was GJK's A).  Out: after the swap the swept link is B.
struct CastOut {
  int cctype;
  float time;
  V3 ptB, ptB1;
  bool between;
};
template <int G>
TOHC_DEV void castClassify(const Cvx<G> &cq, const Xf &tf0, const Xf &tw1, V3 nW, V3 ptA, V3 ptB, bool castFirst, CastOut &o) {
  V3 nFromCast = nW * (-(float)(castFirst ? 1 : -1));
  V3 nl0 = mulVM(nFromCast, tf0.b);
  V3 nl1 = mulVM(nFromCast, tw1.b);
  V3 pl0, pl1;
  float ls0, ls1;
  if constexpr (G >= 2) {
    // every lane of a group would do the same vertex-by-vertex scan twice (the average is order-dependent, so it stays a
    // scan): the lower half of the group takes the first pose's direction, the upper half the second's, in ONE scan
    const bool second = (Grp<G>::lane() & (G / 2)) != 0;
    V3 pl;
    averageSupport<G>(cq, second ? nl1 : nl0, ls0, pl);
    pl0 = mk(Grp<G>::bcast(pl.x, 0), Grp<G>::bcast(pl.y, 0), Grp<G>::bcast(pl.z, 0));
    pl1 = mk(Grp<G>::bcast(pl.x, G / 2), Grp<G>::bcast(pl.y, G / 2), Grp<G>::bcast(pl.z, G / 2));
  } else {
    averageSupport<G>(cq, nl0, ls0, pl0);
    averageSupport<G>(cq, nl1, ls1, pl1);
  }
  V3 pw0 = xfApply(tf0, pl0);
  V3 pw1 = xfApply(tw1, pl1);
  float sup0 = dot(nFromCast, pw0);
  float sup1 = dot(nFromCast, pw1);
  o.between = false;
  if (sup0 - sup1 > .01f) { o.time = 0; o.cctype = 1; }
  else if (sup1 - sup0 > .01f) { o.time = 1; o.cctype = 2; }
  else {
    V3 ptOnCast = castFirst ? ptA : ptB;
    float l0c = len(ptOnCast - pw0), l1c = len(ptOnCast - pw1);
    o.ptB = pw0;
    o.ptB1 = pw1;
    o.cctype = 3;
    o.between = true;
    if (l0c + l1c < .001f) o.time = .5f;
    else o.time = l0c / (l0c + l1c);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ContinuousCheckTrajectory (bullet_collision_checker.cpp:683-753): time of impact of one swept convex geometry
// against one resting geometry by conservative advancement,
//   [ccc] .../NarrowPhaseCollision/btContinuousConvexCollision.cpp:49-241   [tu] LinearMath/btTransformUtil.h:20-140
//   [cs2] .../CollisionShapes/btCollisionShape.cpp:32-58                    [m3] LinearMath/btMatrix3x3.h:448-478,1019-1029
// fp32 in Bullet's operation order.  The reference calls libm's sinf / cosf / acosf here, and the advancement is
// chaotically sensitive to their last bit (a 1-ulp change of one sine moves ~4 % of the fractions visibly: the pair
// detector's degenerate-case branches are discontinuous).  glibc's routines cannot run on the device, so the device uses
// the portable routines below; the CPU oracle and the host build of this file can switch to the same routines, which
// makes GPU-vs-oracle parity bit-exact, while in libm mode they are bit-exact against the reference's Bullet.
// ---------------------------------------------------------------------------------------------------------------
/* Portable float sin / cos / acos: Taylor series in IEEE double (+, *, /, sqrt only, no contraction), rounded to float.
 * The SAME operation sequence is compiled into the device code (tohc_narrow.h) and the CPU oracle (trajopt_oracle.c), so
 * both produce identical bits; accuracy ~1e-16, i.e. the correctly rounded float in all but ~1e-8 of the cases.
 * sin / cos are only called with |x| <= pi/8 (ANGULAR_MOTION_THRESHOLD, btTransformUtil.h:20,58-61). */
TOHC_DEV double ptHorner(const double *c, int n, double x2) {
  double r = c[n - 1];
  for (int k = n - 2; k >= 0; --k) r = c[k] + x2 * r;
  return r;
}
TOHC_DEV float ptSin(float xf) {
  const double c[10] = {1.0, -0.16666666666666666, 0.008333333333333333, -0.0001984126984126984, 2.7557319223985893e-06, -2.505210838544172e-08, 1.6059043836821613e-10, -7.647163731819816e-13, 2.8114572543455206e-15, -8.22063524662433e-18};
  const double x = (double)xf;
  return (float)(x * ptHorner(c, 10, x * x));
}
TOHC_DEV float ptCos(float xf) {
  const double c[10] = {1.0, -0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07, 2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, -1.5619206968586225e-16};
  const double x = (double)xf;
  return (float)ptHorner(c, 10, x * x);
}
TOHC_DEV double ptAsinD(double z) { /* |z| <= ~0.71 */
  const double c[30] = {1.0, 0.16666666666666666, 0.075, 0.044642857142857144, 0.030381944444444444, 0.022372159090909092, 0.017352764423076924, 0.01396484375, 0.011551800896139705, 0.009761609529194078, 0.008390335809616815, 0.0073125258735988454, 0.006447210311889649, 0.005740037670841924, 0.005153309682319905, 0.004660143486915096, 0.004240907093679363, 0.003880964558837669, 0.0035692053938259347, 0.003297059503473485, 0.0030578216492580306, 0.002846178401108942, 0.00265787063820729, 0.0024894486782468836, 0.002338091892111975, 0.0022014739737101384, 0.0020776610325181676, 0.0019650336162772837, 0.0018622264064031275, 0.0017680811205154183};
  return z * ptHorner(c, 30, z * z);
}
TOHC_DEV float ptAcos(float xf) { /* xf in [-1, 1] */
  const double PI = 3.141592653589793;
  const double x = (double)xf;
  if (x > 0.5) return (float)(2.0 * ptAsinD(sqrt((1.0 - x) * 0.5)));
  if (x < -0.5) return (float)(PI - 2.0 * ptAsinD(sqrt((1.0 + x) * 0.5)));
  return (float)(PI * 0.5 - ptAsinD(x));
}
#if defined(__HIPCC__)
TOHC_DEV float libmSin(float x) { return ptSin(x); }
TOHC_DEV float libmCos(float x) { return ptCos(x); }
TOHC_DEV float libmAcos(float x) { return ptAcos(x); }
#else
// host build (tests/emu): 0 = libm's float routines, what the reference calls; 1 = the portable ones the device runs
inline int &emuTrigMode() { static int m = 0; return m; }
TOHC_DEV float libmSin(float x) { return emuTrigMode() ? ptSin(x) : sinf(x); }
TOHC_DEV float libmCos(float x) { return emuTrigMode() ? ptCos(x) : cosf(x); }
TOHC_DEV float libmAcos(float x) { return emuTrigMode() ? ptAcos(x) : acosf(x); }
#endif
struct Q4 { float x, y, z, w; };
TOHC_DEV float m3el(const M3 &m, int i, int j) {
  const V3 r = i == 0 ? m.r[0] : (i == 1 ? m.r[1] : m.r[2]);
  return j == 0 ? r.x : (j == 1 ? r.y : r.z);
}
TOHC_DEV float cofac(const M3 &m, int r1, int c1, int r2, int c2) { return m3el(m, r1, c1) * m3el(m, r2, c2) - m3el(m, r1, c2) * m3el(m, r2, c1); }
TOHC_DEV M3 m3Inverse(const M3 &m) {   // [m3:1019-1029]
  V3 co = mk(cofac(m, 1, 1, 2, 2), cofac(m, 1, 2, 2, 0), cofac(m, 1, 0, 2, 1));
  float det = dot(m.r[0], co);
  float s = 1.0f / det;
  M3 r;
  r.r[0] = mk(co.x * s, cofac(m, 0, 2, 2, 1) * s, cofac(m, 0, 1, 1, 2) * s);
  r.r[1] = mk(co.y * s, cofac(m, 0, 0, 2, 2) * s, cofac(m, 0, 2, 1, 0) * s);
  r.r[2] = mk(co.z * s, cofac(m, 0, 1, 2, 0) * s, cofac(m, 0, 0, 1, 1) * s);
  return r;
}
TOHC_DEV Q4 m3GetRotation(const M3 &m) {   // [m3:448-478], scalar branch
  const float e00 = m.r[0].x, e11 = m.r[1].y, e22 = m.r[2].z;
  float trace = e00 + e11 + e22;
  Q4 q;
  if (trace > 0.0f) {
    float s = sqrtf(trace + 1.0f);
    q.w = (s * 0.5f);
    s = 0.5f / s;
    q.x = ((m.r[2].y - m.r[1].z) * s);
    q.y = ((m.r[0].z - m.r[2].x) * s);
    q.z = ((m.r[1].x - m.r[0].y) * s);
  } else {
    const int i = e00 < e11 ? (e11 < e22 ? 2 : 1) : (e00 < e22 ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    float s = sqrtf(m3el(m, i, i) - m3el(m, j, j) - m3el(m, k, k) + 1.0f);
    const float ti = s * 0.5f;
    s = 0.5f / s;
    const float tw = (m3el(m, k, j) - m3el(m, j, k)) * s;
    const float tj = (m3el(m, j, i) + m3el(m, i, j)) * s;
    const float tk = (m3el(m, k, i) + m3el(m, i, k)) * s;
    q.w = tw;
    q.x = i == 0 ? ti : (j == 0 ? tj : tk);
    q.y = i == 1 ? ti : (j == 1 ? tj : tk);
    q.z = i == 2 ? ti : (j == 2 ? tj : tk);
  }
  return q;
}
TOHC_DEV float q4len2(Q4 q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
TOHC_DEV Q4 q4normalize(Q4 q) {
  const float s = 1.0f / sqrtf(q4len2(q));
  Q4 r;
  r.x = q.x * s; r.y = q.y * s; r.z = q.z * s; r.w = q.w * s;
  return r;
}
TOHC_DEV Q4 q4mul(Q4 a, Q4 b) {
  Q4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
TOHC_DEV M3 m3FromQuat(Q4 q) {   // btMatrix3x3::setRotation, scalar branch
  const float d = q4len2(q);
  const float s = 2.0f / d;
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 r;
  r.r[0] = mk(1.0f - (yy + zz), xy - wz, xz + wy);
  r.r[1] = mk(xy + wz, 1.0f - (xx + zz), yz - wx);
  r.r[2] = mk(xz - wy, yz + wx, 1.0f - (xx + yy));
  return r;
}
// btTransformUtil::calculateVelocity with timeStep 1 [tu:108-139]
TOHC_DEV void calcVelocity(const Xf &t0, const Xf &t1, V3 &linVel, V3 &angVel) {
  linVel = vdiv(t1.o - t0.o, 1.0f);
  const M3 dmat = mulMM(t1.b, m3Inverse(t0.b));
  const Q4 dorn = q4normalize(m3GetRotation(dmat));
  float w = dorn.w;
  if (w < -1.0f) w = -1.0f;
  if (w > 1.0f) w = 1.0f;
  const float angle = 2.0f * libmAcos(w);
  V3 axis = mk(dorn.x, dorn.y, dorn.z);
  const float l2 = len2(axis);
  if (l2 < FLT_EPSILON * FLT_EPSILON) axis = mk(1, 0, 0);
  else axis = vdiv(axis, sqrtf(l2));
  angVel = vdiv(axis * angle, 1.0f);
}
// btTransformUtil::integrateTransform [tu:42-78]
TOHC_DEV Xf integrateTransform(const Xf &cur, V3 linvel, V3 angvel, float timeStep) {
  const float AMT = 0.5f * (6.283185307179586232f * 0.25f);   // ANGULAR_MOTION_THRESHOLD
  Xf out;
  out.o = cur.o + linvel * timeStep;
  V3 axis;
  float fAngle = len(angvel);
  if (fAngle * timeStep > AMT) fAngle = AMT / timeStep;
  if (fAngle < 0.001f) axis = angvel * (0.5f * timeStep - (timeStep * timeStep * timeStep) * (0.020833333333f) * fAngle * fAngle);
  else axis = angvel * (libmSin(0.5f * fAngle * timeStep) / fAngle);
  Q4 dorn;
  dorn.x = axis.x; dorn.y = axis.y; dorn.z = axis.z; dorn.w = libmCos(fAngle * timeStep * 0.5f);
  const Q4 pred = q4normalize(q4mul(dorn, m3GetRotation(cur.b)));
  out.b = m3FromQuat(pred);
  return out;
}
// btCollisionShape::getAngularMotionDisc [cs2:32-58] over getAabb(identity): btTransformAabb with the patched
// SHAPE_EXPANSION (= d); a hull's cached local box still carries the 0.04 construction margin
TOHC_DEV float angularMotionDisc(int kind, V3 dims, V3 lmin, V3 lmax, float d) {
  V3 mn, mx;
  if (kind == KIND_SPHERE) {   // btSphereShape::getAabb: centre +- margin, untouched by the SHAPE_EXPANSION patch
    const float m = dims.x * 1.0f;
    mn = mk(0 - m, 0 - m, 0 - m); mx = mk(0 + m, 0 + m, 0 + m);
  } else if (kind == KIND_HULL) {
    const float CDM = 0.04f;
    const V3 hi = mk(lmax.x + CDM, lmax.y + CDM, lmax.z + CDM), lo = mk(lmin.x - CDM, lmin.y - CDM, lmin.z - CDM);
    V3 he = (hi - lo) * 0.5f;
    he = he + (mk(0, 0, 0) + mk(d, d, d));
    const V3 lc = (hi + lo) * 0.5f;
    const V3 c = mk(1.0f * lc.x + 0.0f * lc.y + 0.0f * lc.z + 0.0f, 0.0f * lc.x + 1.0f * lc.y + 0.0f * lc.z + 0.0f, 0.0f * lc.x + 0.0f * lc.y + 1.0f * lc.z + 0.0f);
    mn = c - he; mx = c + he;
  } else {
    const V3 he = (dims + mk(0, 0, 0)) + mk(d, d, d);
    mn = mk(0, 0, 0) - he; mx = mk(0, 0, 0) + he;
  }
  const float radius = len(mx - mn) * 0.5f;
  const V3 center = (mn + mx) * 0.5f;
  return radius + len(center);
}

// computeClosestPoints [ccc:49-59]: the full pair detector (m_maximumDistanceSquared = BT_LARGE_FLOAT), both stages of
// the penetration solver back to back on the group's EpaMem
template <int G>
TOHC_DEV void closestPoints(EpaMem *mem, const Cvx<G> &A, const Cvx<G> &B, const Xf &tA, const Xf &tB, PairOut &o) {
  GjkState st;
  gjkMain<G>(A, B, tA, tB, 1e18f, st);
  if (st.needEpa) gjkFinishEpa<G>(mem, A, B, tA, tB, st);
  gjkOutput(st, 1e18f, o);
}
struct SweepOut { float fraction; V3 normal, point; };
// calcTimeOfImpact [ccc:90-241] with B at rest + the acceptance test of objectQuerySingleInternal
// (btCollisionWorld.cpp:598-603, without the comparison against the closest hit so far); discA / discB: getAngularMotionDisc
template <int G>
TOHC_DEV bool timeOfImpact(EpaMem *mem, const Cvx<G> &A, float discA, const Xf &fromA, const Xf &toA, const Cvx<G> &B, float discB, const Xf &tB, SweepOut &out) {
  V3 linVelA, angVelA, linVelB, angVelB;
  calcVelocity(fromA, toA, linVelA, angVelA);
  calcVelocity(tB, tB, linVelB, angVelB);
  const float maxAngularProjectedVelocity = len(angVelA) * discA + len(angVelB) * discB;
  const V3 relLinVel = linVelB - linVelA;
  const float relLinVelocLength = len(linVelB - linVelA);
  if ((relLinVelocLength + maxAngularProjectedVelocity) == 0.f) return false;
  float lambda = 0.f, lastLambda = 0.f;
  int numIter = 0;
  const float radius = 0.001f;
  PairOut o;
  closestPoints<G>(mem, A, B, fromA, tB, o);
  if (!o.valid) return false;
  V3 c = o.pointOnB, n = o.normalOnB;
  float dist = o.distance + 0.0f;
  float projectedLinearVelocity = dot(relLinVel, n);
  if ((projectedLinearVelocity + maxAngularProjectedVelocity) <= FLT_EPSILON) return false;
  while (dist > radius) {
    projectedLinearVelocity = dot(relLinVel, n);
    if ((projectedLinearVelocity + maxAngularProjectedVelocity) <= FLT_EPSILON) return false;
    const float dLambda = dist / (projectedLinearVelocity + maxAngularProjectedVelocity);
    lambda = lambda + dLambda;
    if (lambda > 1.0f) return false;
    if (lambda < 0.0f) return false;
    if (lambda <= lastLambda) return false;
    lastLambda = lambda;
    const Xf iA = integrateTransform(fromA, linVelA, angVelA, lambda);
    const Xf iB = integrateTransform(tB, linVelB, angVelB, lambda);
    closestPoints<G>(mem, A, B, iA, iB, o);
    if (!o.valid) return false;
    dist = o.distance + 0.0f;
    c = o.pointOnB;
    n = o.normalOnB;
    numIter++;
    if (numIter > 64) return false;
  }
  if (!(len2(n) > 0.0001f)) return false;
  out.fraction = lambda;
  out.normal = vdiv(n, len(n));   // castResult.m_normal.normalize()
  out.point = c;
  return true;
}


// ---------------------------------------------------------------------------------------------------------------
// RayCastCollision (bullet_collision_checker.cpp:352-360): btSubsimplexConvexCast::calcTimeOfImpact
// [btSubSimplexConvexCast.cpp:38-156] as rayTestSingleInternal calls it (btCollisionWorld.cpp:279-301).  convexA is
// btSphereShape(0) with margin 0: its support vertex is (0,0,0) in every direction; fromA / toA are identity rotations
// at the ray's ends (:856-859); B rests and answers through the VIRTUAL localGetSupportingVertex (margin 0).
// ---------------------------------------------------------------------------------------------------------------
TOHC_DEV V3 interp3(V3 a, V3 b, float rt) {  // btVector3::setInterpolate3, scalar branch (btVector3.h:494-497)
  const float s = 1.0f - rt;
  return mk(s * a.x + rt * b.x, s * a.y + rt * b.y, s * a.z + rt * b.z);
}
TOHC_DEV bool rayConvex(const Cvx<1> &B, const Xf &tB, V3 from, V3 to, float &fraction, V3 &normal) {
  Simplex sx;
  vssReset(sx);
  const V3 linVelA = to - from, linVelB = tB.o - tB.o;
  float lambda = 0.0f;
  V3 oA = from;
  Xf iB = tB;   // interpolatedTransB: only its origin is ever rewritten
  const V3 r = linVelA - linVelB;
  const V3 zero = mk(0, 0, 0);
  V3 supA = zero + oA;
  V3 supB = xfApply(tB, supPlainVirtual<1>(B, mulVM(r, tB.b)));
  V3 v = supA - supB;
  V3 n = mk(0, 0, 0);
  float dist2 = len2(v);
  for (int maxIter = 32; dist2 > 0.0001f && maxIter > 0; --maxIter) {   // while ((dist2 > epsilon) && maxIter--)
    supA = zero + oA;
    supB = xfApply(iB, supPlainVirtual<1>(B, mulVM(v, iB.b)));
    const V3 w = supA - supB;
    const float VdotW = dot(v, w);
    if (lambda > 1.0f) return false;
    if (VdotW > 0.0f) {
      const float VdotR = dot(v, r);
      if (VdotR >= -(FLT_EPSILON * FLT_EPSILON)) return false;
      lambda = lambda - VdotW / VdotR;
      oA = interp3(from, to, lambda);
      iB.o = interp3(tB.o, tB.o, lambda);   // s*o + rt*o: not always o again in fp32
      n = v;
    }
    if (!vssInSimplex(sx, w)) vssAdd(sx, w, supA, supB);
    const bool ok = vssUpdate(sx);   // closest(v): v = m_cachedV whatever the outcome
    v = sx.cV;
    dist2 = ok ? len2(v) : 0.0f;
  }
  fraction = lambda;
  normal = (len2(n) >= (FLT_EPSILON * FLT_EPSILON)) ? vdiv(n, len(n)) : mk(0, 0, 0);
  if (dot(normal, r) >= -0.0f) return false;   // -result.m_allowedPenetration
  return len2(normal) > 0.0001f;               // the acceptance test of rayTestSingleInternal (:304)
}

}  // namespace tohc
