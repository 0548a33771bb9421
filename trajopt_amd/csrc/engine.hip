// engine.hip -- libtrajopt_hipcc.so: the C-ABI of include/trajopt_hipcc.h and the gfx950 kernels behind it.
//
// Pipeline of one batch call (all on the engine's own stream, nothing allocates after warm-up):
//   k_fk            dofs[B*T*K] (fp64)            -> link transforms (fp64 quat+trans) + poses of the robot's objects
//   k_dyn_children  object poses                  -> fp32 3x4 world transform + d-inflated AABB of every moving child
//   k_broadphase    (set, query child, other child) AABB cull with the reference's filters -> compacted pair list
//   k_gjk           one group of G lanes per pair: GJK closest points; separated pairs finish here (contact filter,
//                   cast classification, fused Jacobian epilogue, record write); the rest go to the EPA queue
//   k_epa           GJK2 + EPA (pools in LDS) for queued pairs, then the same epilogue
// Static obstacles are transformed once at upload (the reference re-syncs every object on every call,
// bullet_collision_checker.cpp:629-655).  See DESIGN.md for layouts and rooflines.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/trajopt_hipcc.h"
#include "tohc_narrow.h"

using namespace tohc;

// -----------------------------------------------------------------------------------------------------------------
// device views
// -----------------------------------------------------------------------------------------------------------------
struct DevScene {
  int n_shapes, n_children, n_objects;
  const int *shape_kind, *shape_voff, *shape_vcnt;
  const float *shape_dims, *shape_lmin, *shape_lmax;  // [n][3]
  const float *vx, *vy, *vz;
  const int *child_shape, *child_obj, *child_idx;
  const float *child_localT;                          // [n][12] rows of the 3x3 then origin
  const int *obj_off, *obj_cnt, *obj_compound, *obj_group;
  const unsigned char *allowed;
  const float *stat_xf;                               // [n_children][12] world transform when the object is static
  const float *stat_aabb;                             // [n_children][6]
};
struct DynMap {          // which objects / children move per set (robot links, or everything in "poses" mode)
  const int *obj_dyn;    // [n_objects]  -> slot or -1
  const int *child_dyn;  // [n_children] -> slot or -1
  int n_dyn_obj, n_dyn_children;
};
struct DevRobot {
  int n_links, K, n_joint_inds, affine;
  const int *parent, *jtype, *dof, *dof_col, *link_dyn;  // link_dyn: dyn-object slot of the link's geometry or -1
  const double *T_origin, *axis, *mult, *offs, *q_robot;
  const int *obj_jaclink;                                // [n_objects] link whose chain gives the Jacobian, or -1
  double base_T[7];
  double rot_axis[3];
};
struct PairRec { int set, q, x, pad; };
struct EpaRec { int pair; int isValid; float distance; float n[3]; float pB[3]; };
struct Counters { unsigned int n_pairs, n_epa, overflow, pad; };  // one per enqueue; the record counter is shared so
                                                                   // that several enqueues can append to one slab

struct CallParams {
  int n_cast_sets;   // number of (problem, t) units enumerated
  int T, gap;        // sets per problem; cast: second pose = set + gap (FK mode) ; poses mode: T = 1, gap = 0
  int cast;          // 0 discrete, 1 cast
  int filter_mask;
  int want_jac;
  float d, breaking;
  unsigned int max_pairs, max_records;
};

__device__ __forceinline__ Xf ldXf(const float *p) {
  Xf t;
  t.b.r[0] = mk(p[0], p[1], p[2]);
  t.b.r[1] = mk(p[3], p[4], p[5]);
  t.b.r[2] = mk(p[6], p[7], p[8]);
  t.o = mk(p[9], p[10], p[11]);
  return t;
}
__host__ __device__ __forceinline__ void stXf(float *p, const Xf &t) {
  for (int i = 0; i < 3; ++i) { p[3 * i] = t.b.r[i].x; p[3 * i + 1] = t.b.r[i].y; p[3 * i + 2] = t.b.r[i].z; }
  p[9] = t.o.x; p[10] = t.o.y; p[11] = t.o.z;
}
__host__ __device__ __forceinline__ Xf ldXfH(const float *p) {
  Xf t;
  t.b.r[0] = mk(p[0], p[1], p[2]);
  t.b.r[1] = mk(p[3], p[4], p[5]);
  t.b.r[2] = mk(p[6], p[7], p[8]);
  t.o = mk(p[9], p[10], p[11]);
  return t;
}

// -----------------------------------------------------------------------------------------------------------------
// k_fk: RobotAndDOF::SetDOFValues (configuration_space.cpp:12-23) + link->GetTransform() for every link.
// One thread per configuration; the chain is walked parents-first, each link reading its parent's result back.
// -----------------------------------------------------------------------------------------------------------------
__global__ void k_fk(DevRobot rb, const double *__restrict__ dofs, int n_cfg, double *__restrict__ linkQT, double *__restrict__ objQT,
                     int n_dyn_obj) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_cfg) return;
  const double *q = dofs + (size_t)s * rb.K;
  Dxf base = dxfFrom7(rb.base_T);
  if (rb.affine) {
    const double *a = q + rb.n_joint_inds;
    base.q.w = 1; base.q.x = base.q.y = base.q.z = 0;
    base.t = mkd(0, 0, 0);
    if (rb.affine & TOHC_DOF_X) base.t.x = *a++;
    if (rb.affine & TOHC_DOF_Y) base.t.y = *a++;
    if (rb.affine & TOHC_DOF_Z) base.t.z = *a++;
    if (rb.affine & TOHC_DOF_ROTAXIS) {
      double h = 0.5 * (*a++), sn = sin(h);
      base.q.w = cos(h);
      base.q.x = rb.rot_axis[0] * sn; base.q.y = rb.rot_axis[1] * sn; base.q.z = rb.rot_axis[2] * sn;
    }
  }
  double *out = linkQT + (size_t)s * rb.n_links * 7;
  for (int i = 0; i < rb.n_links; ++i) {
    Dxf parent = rb.parent[i] < 0 ? base : dxfFrom7(out + 7 * rb.parent[i]);
    Dxf t = dxfMul(parent, dxfFrom7(rb.T_origin + 7 * i));
    if (rb.jtype[i] != TOHC_J_FIXED) {
      double v = rb.offs[i];
      if (rb.dof[i] >= 0) {
        double qv = rb.dof_col[i] >= 0 ? q[rb.dof_col[i]] : rb.q_robot[rb.dof[i]];
        v = rb.mult[i] * qv + rb.offs[i];
      }
      D3 ax = mkd(rb.axis[3 * i], rb.axis[3 * i + 1], rb.axis[3 * i + 2]);
      Dxf m;
      m.q.w = 1; m.q.x = m.q.y = m.q.z = 0;
      m.t = mkd(0, 0, 0);
      if (rb.jtype[i] == TOHC_J_REVOLUTE) {
        double h = 0.5 * v, sn = sin(h);
        m.q.w = cos(h); m.q.x = ax.x * sn; m.q.y = ax.y * sn; m.q.z = ax.z * sn;
      } else {
        m.t = ax * v;
      }
      t = dxfMul(t, m);
    }
    dxfTo7(t, out + 7 * i);
    if (objQT && rb.link_dyn[i] >= 0) dxfTo7(t, objQT + ((size_t)s * n_dyn_obj + rb.link_dyn[i]) * 7);
  }
}

// k_dyn_children: object pose (7 doubles) -> fp32 transform exactly as toBt() does, child world transform
// (orgTrans*childTrans for compounds, btCompoundCollisionAlgorithm.cpp:122-125) and the d-inflated cull box.
__global__ void k_dyn_children(DevScene sc, DynMap dm, const int *__restrict__ dyn_child_list, const double *__restrict__ objQT,
                               int n_sets, float d, float *__restrict__ dynXf, float *__restrict__ dynAabb) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)n_sets * dm.n_dyn_children;
  if (i >= total) return;
  int set = (int)(i / dm.n_dyn_children), slot = (int)(i % dm.n_dyn_children);
  int c = dyn_child_list[slot];
  int obj = sc.child_obj[c];
  Xf T = xfFrom7(objQT + ((size_t)set * dm.n_dyn_obj + dm.obj_dyn[obj]) * 7);
  if (sc.obj_compound[obj]) T = xfMul(T, ldXf(sc.child_localT + 12 * c));
  stXf(dynXf + i * 12, T);
  int sh = sc.child_shape[c];
  Aabb b = shapeAabb(mk(sc.shape_lmin[3 * sh], sc.shape_lmin[3 * sh + 1], sc.shape_lmin[3 * sh + 2]),
                     mk(sc.shape_lmax[3 * sh], sc.shape_lmax[3 * sh + 1], sc.shape_lmax[3 * sh + 2]), T, d);
  float *o = dynAabb + i * 6;
  o[0] = b.lo.x; o[1] = b.lo.y; o[2] = b.lo.z; o[3] = b.hi.x; o[4] = b.hi.y; o[5] = b.hi.z;
}

__device__ __forceinline__ Aabb ldAabb(const float *p) {
  Aabb b;
  b.lo = mk(p[0], p[1], p[2]);
  b.hi = mk(p[3], p[4], p[5]);
  return b;
}
__device__ __forceinline__ void castSets(const CallParams &cp, int cs, int &set0, int &set1) {
  int per = cp.cast ? (cp.T - cp.gap) : cp.T;
  set0 = (cs / per) * cp.T + cs % per;
  set1 = set0 + (cp.cast ? cp.gap : 0);
}

// -----------------------------------------------------------------------------------------------------------------
// k_broadphase: one thread per (unit, query child, other child).  Filters follow CollisionCollector::needsCollision
// (bullet_collision_checker.cpp:409-413) and btSingleContactCallback::process (btCollisionWorld.cpp:1129-1137); the
// cull is the engine's own conservative box test (see oracle/trajopt_oracle.c, shapeAabb): both boxes grown by d,
// closed-interval overlap -- it can never drop a pair whose distance is <= d.
// -----------------------------------------------------------------------------------------------------------------
__global__ void k_broadphase(DevScene sc, DynMap dm, CallParams cp, const int *__restrict__ qchildren, int nQ,
                             const float *__restrict__ dynAabb0, const float *__restrict__ dynAabb1, PairRec *__restrict__ pairs,
                             Counters *__restrict__ cnt) {
  const int xTiles = (sc.n_children + blockDim.x - 1) / blockDim.x;
  const size_t blk = blockIdx.x;
  const int tile = (int)(blk % xTiles);
  const size_t uq = blk / xTiles;
  const int qi = (int)(uq % nQ);
  const int cs = (int)(uq / nQ);
  const int x = tile * blockDim.x + threadIdx.x;
  if (x >= sc.n_children) return;
  const int q = qchildren[qi];
  const int Q = sc.child_obj[q], X = sc.child_obj[x];
  if (Q == X) return;
  const int gx = sc.obj_group[X];
  if (cp.cast) {
    if (!(gx & TOHC_GROUP_KINBODY)) return;  // cc.m_collisionFilterMask = KinBodyFilter (bullet_collision_checker.cpp:983)
  } else {
    if (!(gx & (short)cp.filter_mask)) return;
  }
  if (!sc.allowed[(size_t)Q * sc.n_objects + X]) return;
  int set0, set1;
  castSets(cp, cs, set0, set1);
  const int qs = dm.child_dyn[q];
  Aabb bq = qs >= 0 ? ldAabb(dynAabb0 + ((size_t)set0 * dm.n_dyn_children + qs) * 6) : ldAabb(sc.stat_aabb + 6 * q);
  if (cp.cast && qs >= 0) bq = aabbUnion(bq, ldAabb(dynAabb1 + ((size_t)set1 * dm.n_dyn_children + qs) * 6));
  const int xs = dm.child_dyn[x];
  Aabb bx = xs >= 0 ? ldAabb(dynAabb0 + ((size_t)set0 * dm.n_dyn_children + xs) * 6) : ldAabb(sc.stat_aabb + 6 * x);
  if (!aabbOverlap(bq, bx)) return;
  unsigned int slot = atomicAdd(&cnt->n_pairs, 1u);
  if (slot < cp.max_pairs) {
    PairRec p;
    p.set = cs; p.q = q; p.x = x; p.pad = 0;
    pairs[slot] = p;
  } else {
    cnt->overflow = 1;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// narrowphase
// -----------------------------------------------------------------------------------------------------------------
struct NarrowArgs {
  DevScene sc;
  DynMap dm;
  DevRobot rb;
  CallParams cp;
  const float *dynXf0, *dynXf1;
  const double *linkQT;   // [n_sets][n_links][7], null in poses mode
  const double *dofs;     // [n_sets][K]
  const PairRec *pairs;
  EpaRec *epaq;
  Counters *cnt;
  unsigned int *n_records;
  tohc_record *records;
  double *grads;          // [max_records][2][K]
};

template <int G>
struct PairCtx {
  Cvx<G> A, B;
  Xf tA, tB;
  // bookkeeping for the epilogue
  int set0, set1, Q, qc, X, xc;
  bool xIsA;       // X's child is GJK's A (X compound), else the query child is A
  Xf tf0, tw1;     // cast: pose of the swept child at both ends
};

template <int G>
__device__ __forceinline__ void loadShape(const DevScene &sc, int child, Cvx<G> &c) {
  const int sh = sc.child_shape[child];
  cvxLoad<G>(c, sc.shape_kind[sh], mk(sc.shape_dims[3 * sh], sc.shape_dims[3 * sh + 1], sc.shape_dims[3 * sh + 2]), sc.vx, sc.vy, sc.vz,
             sc.shape_voff[sh], sc.shape_vcnt[sh]);
}
__device__ __forceinline__ Xf childXf(const NarrowArgs &a, const float *dyn, int set, int child) {
  const int s = a.dm.child_dyn[child];
  return s >= 0 ? ldXf(dyn + ((size_t)set * a.dm.n_dyn_children + s) * 12) : ldXf(a.sc.stat_xf + 12 * child);
}

template <int G>
__device__ __forceinline__ void setupPair(const NarrowArgs &a, const PairRec &p, PairCtx<G> &c) {
  castSets(a.cp, p.set, c.set0, c.set1);
  c.Q = a.sc.child_obj[p.q]; c.X = a.sc.child_obj[p.x];
  c.qc = a.sc.child_idx[p.q]; c.xc = a.sc.child_idx[p.x];
  c.xIsA = a.sc.obj_compound[c.X] != 0;
  Xf tq = childXf(a, a.dynXf0, c.set0, p.q);
  Xf tx = childXf(a, a.dynXf0, c.set0, p.x);
  Cvx<G> &cq = c.xIsA ? c.B : c.A;
  Cvx<G> &cx = c.xIsA ? c.A : c.B;
  loadShape<G>(a.sc, p.q, cq);
  loadShape<G>(a.sc, p.x, cx);
  if (a.cp.cast) {
    Xf tq1 = childXf(a, a.dynXf1, c.set1, p.q);
    cq.cast = 1;
    cq.t01 = xfInverseTimes(tq, tq1);          // CastHullShape(convex, tf0.inverseTimes(tf1)), bullet_collision_checker.cpp:977
    c.tf0 = tq;
    c.tw1 = xfMul(tq, cq.t01);                 // m_cow->getWorldTransform() * shape->m_t01, :904
  }
  if (c.xIsA) { c.tA = tx; c.tB = tq; } else { c.tA = tq; c.tB = tx; }
}

// point Jacobian of `link` at world point p, column k: CalculateActiveJacobian as used by PositionJacobian
// (configuration_space.cpp:83-89).  Chain walked leaf -> root like the oracle, so sums round identically.
__device__ __forceinline__ D3 jacColumn(const DevRobot &rb, const double *linkQT, const double *q, int link, D3 p, int k) {
  D3 J = mkd(0, 0, 0);
  if (k < rb.n_joint_inds) {
    for (int i = link; i >= 0; i = rb.parent[i]) {
      if (rb.jtype[i] == TOHC_J_FIXED || rb.dof_col[i] != k) continue;
      Dxf t = dxfFrom7(linkQT + 7 * i);
      D3 z = qrot(t.q, mkd(rb.axis[3 * i], rb.axis[3 * i + 1], rb.axis[3 * i + 2]));
      D3 col = rb.jtype[i] == TOHC_J_REVOLUTE ? cross(z, p - t.t) : z;
      J.x += rb.mult[i] * col.x; J.y += rb.mult[i] * col.y; J.z += rb.mult[i] * col.z;
    }
    return J;
  }
  int kk = rb.n_joint_inds;
  if (rb.affine & TOHC_DOF_X) { if (k == kk) return mkd(1, 0, 0); ++kk; }
  if (rb.affine & TOHC_DOF_Y) { if (k == kk) return mkd(0, 1, 0); ++kk; }
  if (rb.affine & TOHC_DOF_Z) { if (k == kk) return mkd(0, 0, 1); ++kk; }
  if (rb.affine & TOHC_DOF_ROTAXIS) {
    if (k == kk) {
      const double *a = q + rb.n_joint_inds;
      D3 bt = mkd(0, 0, 0);
      if (rb.affine & TOHC_DOF_X) bt.x = *a++;
      if (rb.affine & TOHC_DOF_Y) bt.y = *a++;
      if (rb.affine & TOHC_DOF_Z) bt.z = *a++;
      return cross(mkd(rb.rot_axis[0], rb.rot_axis[1], rb.rot_axis[2]), p - bt);
    }
  }
  return J;
}

// CollisionsToDistanceExpressions for one contact at one configuration (collision_terms.cpp:44-56):
// lanes of the group take the columns k = gl, gl+G, ...; the constant term is summed in column order.
template <int G>
__device__ __forceinline__ double linearizeAt(const NarrowArgs &a, int set, int la, int lb, D3 n, D3 ptA, D3 ptB, double dist, double *gradOut) {
  const DevRobot &rb = a.rb;
  const int gl = Grp<G>::lane();
  const double *linkQT = a.linkQT + (size_t)set * rb.n_links * 7;
  const double *q = a.dofs + (size_t)set * rb.K;
  double gq = 0;
  for (int k0 = 0; k0 < rb.K; k0 += G) {
    const int k = k0 + gl;
    double g = 0;
    if (k < rb.K) {
      if (la >= 0) {
        D3 J = jacColumn(rb, linkQT, q, la, ptA, k);
        g += 1.0 * (n.x * J.x + n.y * J.y + n.z * J.z);
      }
      if (lb >= 0) {
        D3 J = jacColumn(rb, linkQT, q, lb, ptB, k);
        g += -1.0 * (n.x * J.x + n.y * J.y + n.z * J.z);
      }
      gradOut[k] = g;
    }
    const int lim = min(G, rb.K - k0);
    for (int j = 0; j < lim; ++j) {
      double gj = __shfl(g, j, G);
      gq += gj * q[k0 + j];
    }
  }
  return dist - gq;
}

// Everything after the narrowphase for one pair: Bullet's output filter, the collector's distance filter, the cast
// classification, the record and the fused Jacobian rows.
template <int G>
__device__ __forceinline__ void finishPair(const NarrowArgs &a, const PairCtx<G> &c, const GjkState &st) {
  float md2 = a.cp.breaking;
  md2 *= md2;
  PairOut o;
  gjkOutput(st, md2, o);
  if (!o.valid) return;
  if ((double)o.distance > (double)a.cp.d) return;  // CollisionCollector::addSingleResult, bullet_collision_checker.cpp:401
  // btBridgedManifoldResult::addContactPoint (btCollisionWorld.cpp:1068-1086)
  V3 ptA = o.pointOnB + o.normalOnB * o.distance;
  V3 ptB = o.pointOnB;
  V3 nrm = o.normalOnB;
  int idA = c.xIsA ? c.X : c.Q, chA = c.xIsA ? c.xc : c.qc;
  int idB = c.xIsA ? c.Q : c.X, chB = c.xIsA ? c.qc : c.xc;
  int cctype = TOHC_CC_NONE;
  float time = 0;
  V3 ptB1 = mk(0, 0, 0);
  if (a.cp.cast) {
    const bool castFirst = !c.xIsA;
    const Cvx<G> &cq = c.xIsA ? c.B : c.A;
    CastOut co;
    castClassify<G>(cq, c.tf0, c.tw1, nrm, ptA, ptB, castFirst, co);
    if (castFirst) {  // swap so that the swept link is B (bullet_collision_checker.cpp:913-917)
      V3 t = ptA; ptA = ptB; ptB = t;
      nrm = nrm * -1.0f;
      int ti = idA; idA = idB; idB = ti;
      ti = chA; chA = chB; chB = ti;
    }
    cctype = co.cctype;
    time = co.time;
    if (co.between) { ptB = co.ptB; ptB1 = co.ptB1; }
  }
  const int gl = Grp<G>::lane();
  unsigned int slot = 0;
  if (gl == 0) slot = atomicAdd(a.n_records, 1u);
  slot = (unsigned int)__shfl((int)slot, 0, G);
  if (slot >= a.cp.max_records) {
    if (gl == 0) a.cnt->overflow = 1;
    return;
  }
  double c0 = 0, c1 = 0;
  if (a.cp.want_jac && a.linkQT) {
    const int la = a.rb.obj_jaclink[idA], lb = a.rb.obj_jaclink[idB];
    D3 n = mkd(nrm.x, nrm.y, nrm.z), pA = mkd(ptA.x, ptA.y, ptA.z), pB = mkd(ptB.x, ptB.y, ptB.z), pB1 = mkd(ptB1.x, ptB1.y, ptB1.z);
    double *g = a.grads + (size_t)slot * 2 * a.rb.K;
    c0 = linearizeAt<G>(a, c.set0, la, lb, n, pA, pB, (double)o.distance, g);
    if (a.cp.cast) {
      c1 = linearizeAt<G>(a, c.set1, la, lb, n, pA, (cctype == TOHC_CC_BETWEEN) ? pB1 : pB, (double)o.distance, g + a.rb.K);
    } else {
      for (int k = gl; k < a.rb.K; k += G) g[a.rb.K + k] = 0;
    }
  }
  if (gl == 0) {
    tohc_record r;
    r.problem = c.set0 / a.cp.T;
    r.t = c.set0 % a.cp.T;
    r.idA = idA; r.childA = chA; r.idB = idB; r.childB = chB;
    r.cctype = cctype;
    r.time = time;
    r.dist = o.distance;
    r.n[0] = nrm.x; r.n[1] = nrm.y; r.n[2] = nrm.z;
    r.ptA[0] = ptA.x; r.ptA[1] = ptA.y; r.ptA[2] = ptA.z;
    r.ptB[0] = ptB.x; r.ptB[1] = ptB.y; r.ptB[2] = ptB.z;
    r.ptB1[0] = ptB1.x; r.ptB1[1] = ptB1.y; r.ptB1[2] = ptB1.z;
    r.const0 = c0; r.const1 = c1;
    a.records[slot] = r;
  }
}

template <int G>
__global__ void __launch_bounds__(256) k_gjk(NarrowArgs a) {
  const int groupsPerBlock = blockDim.x / G;
  const unsigned int gid = blockIdx.x * groupsPerBlock + threadIdx.x / G;
  const unsigned int ngroups = gridDim.x * groupsPerBlock;
  unsigned int npairs = a.cnt->n_pairs;
  if (npairs > a.cp.max_pairs) npairs = a.cp.max_pairs;
  const int gl = Grp<G>::lane();
  float md2 = a.cp.breaking;
  md2 *= md2;
  for (unsigned int p = gid; p < npairs; p += ngroups) {
    PairCtx<G> c;
    setupPair<G>(a, a.pairs[p], c);
    GjkState st;
    gjkMain<G>(c.A, c.B, c.tA, c.tB, md2, st);
    if (st.needEpa) {
      unsigned int slot = 0;
      if (gl == 0) {
        slot = atomicAdd(&a.cnt->n_epa, 1u);
        EpaRec e;
        e.pair = (int)p;
        e.isValid = st.isValid;
        e.distance = st.distance;
        e.n[0] = st.normalInB.x; e.n[1] = st.normalInB.y; e.n[2] = st.normalInB.z;
        e.pB[0] = st.pointOnB.x; e.pB[1] = st.pointOnB.y; e.pB[2] = st.pointOnB.z;
        a.epaq[slot] = e;   // the queue is as long as the pair list
      }
    } else {
      finishPair<G>(a, c, st);
    }
  }
}

template <int G, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_epa(NarrowArgs a) {
  __shared__ EpaMem mem[BLOCK / G];
  const int groupsPerBlock = BLOCK / G;
  const unsigned int gid = blockIdx.x * groupsPerBlock + threadIdx.x / G;
  const unsigned int ngroups = gridDim.x * groupsPerBlock;
  unsigned int nepa = a.cnt->n_epa;
  if (nepa > a.cp.max_pairs) nepa = a.cp.max_pairs;
  EpaMem *m = &mem[threadIdx.x / G];
  for (unsigned int i = gid; i < nepa; i += ngroups) {
    const EpaRec e = a.epaq[i];
    PairCtx<G> c;
    setupPair<G>(a, a.pairs[e.pair], c);
    GjkState st;
    st.isValid = e.isValid != 0;
    st.needEpa = true;
    st.distance = e.distance;
    st.normalInB = mk(e.n[0], e.n[1], e.n[2]);
    st.pointOnB = mk(e.pB[0], e.pB[1], e.pB[2]);
    st.offset = (c.tA.o + c.tB.o) * 0.5f;
    st.iters = 0;
    gjkFinishEpa<G>(m, c.A, c.B, c.tA, c.tB, st);
    finishPair<G>(a, c, st);
  }
}

// -----------------------------------------------------------------------------------------------------------------
// host side
// -----------------------------------------------------------------------------------------------------------------
template <class T>
struct DBuf {  // grow-only device buffer
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <class T>
struct HBuf {  // grow-only pinned host buffer
  T *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipHostMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct tohc_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  int group = 16;

  // scene (host copies kept for static transforms and re-inflation)
  bool have_scene = false, have_robot = false;
  int n_shapes = 0, n_children = 0, n_objects = 0, n_verts = 0;
  std::vector<int> h_shape_kind, h_shape_voff, h_shape_vcnt, h_child_shape, h_child_obj, h_child_idx;
  std::vector<float> h_shape_dims, h_shape_lmin, h_shape_lmax, h_child_localT;
  std::vector<int> h_obj_off, h_obj_cnt, h_obj_compound, h_obj_group;
  std::vector<double> h_obj_static_T;
  std::vector<unsigned char> h_allowed;
  DBuf<int> d_shape_kind, d_shape_voff, d_shape_vcnt, d_child_shape, d_child_obj, d_child_idx, d_obj_off, d_obj_cnt, d_obj_compound, d_obj_group;
  DBuf<float> d_shape_dims, d_shape_lmin, d_shape_lmax, d_vx, d_vy, d_vz, d_child_localT, d_stat_xf, d_stat_aabb;
  DBuf<unsigned char> d_allowed;
  float d = 0.05f, breaking = 0;

  // robot
  int n_links = 0, K = 0, n_joint_inds = 0, affine = 0, n_dof = 0;
  std::vector<int> h_parent, h_jtype, h_dof, h_dof_col, h_link_obj, h_link_dyn, h_obj_jaclink;
  double base_T[7], rot_axis[3];
  DBuf<int> d_parent, d_jtype, d_dof, d_dof_col, d_link_dyn, d_obj_jaclink;
  DBuf<double> d_T_origin, d_axis, d_mult, d_offs, d_q_robot;
  // dynamic maps: robot mode and all-objects ("poses") mode
  std::vector<int> h_obj_dynR, h_child_dynR, h_dyn_childrenR, h_qchildrenR;
  std::vector<int> h_obj_dynP, h_child_dynP, h_dyn_childrenP;
  DBuf<int> d_obj_dynR, d_child_dynR, d_dyn_childrenR, d_qchildrenR, d_obj_dynP, d_child_dynP, d_dyn_childrenP, d_qchildrenP;
  int n_dyn_objR = 0, n_dyn_childrenR = 0;

  // per-call buffers
  DBuf<double> d_dofs, d_linkQT, d_objQT, d_objQT1, d_grads;
  DBuf<float> d_dynXf, d_dynAabb, d_dynXf1, d_dynAabb1;
  DBuf<PairRec> d_pairs;
  DBuf<EpaRec> d_epaq;
  DBuf<Counters> d_cnt;            // [RING]
  DBuf<tohc_record> d_records;
  HBuf<tohc_record> h_records, h_sorted;
  HBuf<double> h_grads, h_gsorted;
  HBuf<Counters> h_cnt;            // [RING]
  size_t max_pairs = (size_t)1 << 22, max_records = (size_t)1 << 18;
  tohc_counters counters;
  // enqueue ring: every enqueue since the last sync has its own device counter slot and event set
  struct Pending { int slot; int64_t aabb_tests; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> events;   // RING * 6
  DBuf<unsigned int> d_nrec;        // shared record counter
  HBuf<unsigned int> h_nrec;
  int staged_B = 0, staged_T = 0;
  bool staged = false;
  bool last_robot_mode = true, last_want_jac = false;
  size_t fetched_n = 0;
  tohc_engine() { memset(&counters, 0, sizeof counters); }
};
static const int RING = 32;

static int fail(tohc_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}
#define HIPCHK(call)                                                                                         \
  do {                                                                                                       \
    hipError_t _e = (call);                                                                                  \
    if (_e != hipSuccess) return fail(e, TOHC_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

template <class T>
static hipError_t upload(DBuf<T> &b, const T *src, size_t n, hipStream_t s) {
  hipError_t e = b.reserve(n);
  if (e != hipSuccess) return e;
  if (n == 0) return hipSuccess;
  return hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
}
template <class T>
static hipError_t upload(DBuf<T> &b, const std::vector<T> &v, hipStream_t s) { return upload(b, v.data(), v.size(), s); }

static float safeMargin(float cur, float x, float y, float z) {  // btConvexInternalShape::setSafeMargin, btConvexInternalShape.h:70-85
  float mn = x < y ? (x < z ? x : z) : (y < z ? y : z);
  float safe = 0.1f * mn;
  return safe < cur ? safe : cur;
}

extern "C" {

int tohc_create(tohc_engine **out, int device) {
  if (!out) return TOHC_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TOHC_E_HIP;  // no CPU fallback: fail loudly
  if (device < 0 || device >= ndev) return TOHC_E_INVALID;
  tohc_engine *e = new tohc_engine;
  e->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) {
    delete e;
    return TOHC_E_HIP;
  }
  const char *g = getenv("TOHC_GROUP");
  if (g && atoi(g) == 64) e->group = 64;
  e->breaking = (float)(2.001 * (double)e->d);
  e->events.resize((size_t)RING * 6);
  for (auto &ev : e->events)
    if (hipEventCreate(&ev) != hipSuccess) { delete e; return TOHC_E_HIP; }
  if (e->d_cnt.reserve(RING) != hipSuccess || e->h_cnt.reserve(RING) != hipSuccess || e->d_nrec.reserve(1) != hipSuccess ||
      e->h_nrec.reserve(1) != hipSuccess || hipMemset(e->d_nrec.p, 0, sizeof(unsigned int)) != hipSuccess) { delete e; return TOHC_E_HIP; }
  *out = e;
  return TOHC_OK;
}

void tohc_destroy(tohc_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  DBuf<int> *ib[] = {&e->d_shape_kind, &e->d_shape_voff, &e->d_shape_vcnt, &e->d_child_shape, &e->d_child_obj, &e->d_child_idx, &e->d_obj_off,
                     &e->d_obj_cnt, &e->d_obj_compound, &e->d_obj_group, &e->d_parent, &e->d_jtype, &e->d_dof, &e->d_dof_col, &e->d_link_dyn,
                     &e->d_obj_jaclink, &e->d_obj_dynR, &e->d_child_dynR, &e->d_dyn_childrenR, &e->d_qchildrenR, &e->d_obj_dynP,
                     &e->d_child_dynP, &e->d_dyn_childrenP, &e->d_qchildrenP};
  for (auto b : ib) b->release();
  DBuf<float> *fb[] = {&e->d_shape_dims, &e->d_shape_lmin, &e->d_shape_lmax, &e->d_vx, &e->d_vy, &e->d_vz, &e->d_child_localT, &e->d_stat_xf,
                       &e->d_stat_aabb, &e->d_dynXf, &e->d_dynAabb, &e->d_dynXf1, &e->d_dynAabb1};
  for (auto b : fb) b->release();
  DBuf<double> *db[] = {&e->d_T_origin, &e->d_axis, &e->d_mult, &e->d_offs, &e->d_q_robot, &e->d_dofs, &e->d_linkQT, &e->d_objQT, &e->d_objQT1, &e->d_grads};
  for (auto b : db) b->release();
  e->d_allowed.release(); e->d_pairs.release(); e->d_epaq.release(); e->d_cnt.release(); e->d_records.release();
  e->h_records.release(); e->h_sorted.release(); e->h_grads.release(); e->h_gsorted.release(); e->h_cnt.release();
  (void)hipEventDestroy(e->ev0); (void)hipEventDestroy(e->ev1);
  for (auto ev : e->events) (void)hipEventDestroy(ev);
  e->d_nrec.release(); e->h_nrec.release();
  (void)hipStreamDestroy(e->stream);
  delete e;
}

const char *tohc_last_error(const tohc_engine *e) { return e ? e->err.c_str() : "null engine"; }

static int refreshStatic(tohc_engine *e) {
  // world transform + d-inflated box of every child as if its object were static (used when it is not dynamic)
  std::vector<float> xf((size_t)e->n_children * 12), bb((size_t)e->n_children * 6);
  for (int c = 0; c < e->n_children; ++c) {
    int obj = e->h_child_obj[c];
    Xf T = xfFrom7(&e->h_obj_static_T[7 * obj]);
    if (e->h_obj_compound[obj]) T = xfMul(T, ldXfH(&e->h_child_localT[12 * c]));
    stXf(&xf[12 * c], T);
    int sh = e->h_child_shape[c];
    Aabb b = shapeAabb(mk(e->h_shape_lmin[3 * sh], e->h_shape_lmin[3 * sh + 1], e->h_shape_lmin[3 * sh + 2]),
                       mk(e->h_shape_lmax[3 * sh], e->h_shape_lmax[3 * sh + 1], e->h_shape_lmax[3 * sh + 2]), T, e->d);
    float *o = &bb[6 * c];
    o[0] = b.lo.x; o[1] = b.lo.y; o[2] = b.lo.z; o[3] = b.hi.x; o[4] = b.hi.y; o[5] = b.hi.z;
  }
  HIPCHK(upload(e->d_stat_xf, xf, e->stream));
  HIPCHK(upload(e->d_stat_aabb, bb, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return TOHC_OK;
}

int tohc_upload_scene(tohc_engine *e, int n_shapes, const int32_t *shape_kind, const float *shape_param, const int32_t *shape_voff,
                      const int32_t *shape_vcnt, int n_verts, const float *vx, const float *vy, const float *vz, int n_children,
                      const int32_t *child_shape, const double *child_T, int n_objects, const int32_t *obj_child_off,
                      const int32_t *obj_child_cnt, const int32_t *obj_compound, const int32_t *obj_group, const double *obj_static_T) {
  if (!e) return TOHC_E_INVALID;
  if (n_shapes <= 0 || n_children <= 0 || n_objects <= 0 || !shape_kind || !shape_param || !child_shape || !child_T || !obj_child_off ||
      !obj_child_cnt || !obj_compound || !obj_group || !obj_static_T)
    return fail(e, TOHC_E_INVALID, "upload_scene: null or empty input");
  HIPCHK(hipSetDevice(e->device));
  e->n_shapes = n_shapes; e->n_children = n_children; e->n_objects = n_objects; e->n_verts = n_verts;
  e->h_shape_kind.assign(shape_kind, shape_kind + n_shapes);
  e->h_shape_voff.assign(n_shapes, 0); e->h_shape_vcnt.assign(n_shapes, 0);
  e->h_shape_dims.assign((size_t)n_shapes * 3, 0.f); e->h_shape_lmin.assign((size_t)n_shapes * 3, 0.f); e->h_shape_lmax.assign((size_t)n_shapes * 3, 0.f);
  for (int i = 0; i < n_shapes; ++i) {
    const float *p = shape_param + 4 * i;
    if (shape_kind[i] == TOHC_BOX) {
      // btBoxShape ctor (setSafeMargin; dims = h - margin) followed by the checker's setMargin(0): (dims + margin) - 0
      // (btBoxShape.cpp:17-26, btBoxShape.h:87-96, bullet_collision_checker.cpp:227,244)
      if (!(p[0] > 0 && p[1] > 0 && p[2] > 0)) return fail(e, TOHC_E_INVALID, "upload_scene: box %d has non-positive extents", i);
      float m = safeMargin(0.04f, p[0], p[1], p[2]);
      for (int k = 0; k < 3; ++k) {
        float dim = ((p[k] * 1.0f - m) + m) - 0.0f;
        e->h_shape_dims[3 * i + k] = dim; e->h_shape_lmin[3 * i + k] = -dim; e->h_shape_lmax[3 * i + k] = dim;
      }
    } else if (shape_kind[i] == TOHC_HULL) {
      int off = shape_voff[i], cnt = shape_vcnt[i];
      if (!vx || !vy || !vz || cnt < 1 || cnt > 64 || off < 0 || off + cnt > n_verts)
        return fail(e, TOHC_E_INVALID, "upload_scene: hull %d has a bad vertex range (1..64 vertices supported)", i);
      e->h_shape_voff[i] = off; e->h_shape_vcnt[i] = cnt;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int k = 0; k < cnt; ++k) {
        float q[3] = {vx[off + k], vy[off + k], vz[off + k]};
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], q[a]); hi[a] = fmaxf(hi[a], q[a]); }
      }
      for (int a = 0; a < 3; ++a) { e->h_shape_lmin[3 * i + a] = lo[a]; e->h_shape_lmax[3 * i + a] = hi[a]; }
    } else {
      return fail(e, TOHC_E_UNSUPPORTED, "upload_scene: shape %d has kind %d; only boxes and convex hulls are supported", i, shape_kind[i]);
    }
  }
  e->h_child_shape.assign(child_shape, child_shape + n_children);
  e->h_child_obj.assign(n_children, -1); e->h_child_idx.assign(n_children, 0);
  e->h_child_localT.assign((size_t)n_children * 12, 0.f);
  for (int c = 0; c < n_children; ++c) {
    if (child_shape[c] < 0 || child_shape[c] >= n_shapes) return fail(e, TOHC_E_INVALID, "upload_scene: child %d references shape %d", c, child_shape[c]);
    stXf(&e->h_child_localT[12 * c], xfFrom7(child_T + 7 * c));
  }
  e->h_obj_off.assign(obj_child_off, obj_child_off + n_objects); e->h_obj_cnt.assign(obj_child_cnt, obj_child_cnt + n_objects);
  e->h_obj_compound.assign(obj_compound, obj_compound + n_objects); e->h_obj_group.assign(obj_group, obj_group + n_objects);
  e->h_obj_static_T.assign(obj_static_T, obj_static_T + (size_t)7 * n_objects);
  for (int o = 0; o < n_objects; ++o) {
    if (obj_child_off[o] < 0 || obj_child_cnt[o] < 1 || obj_child_off[o] + obj_child_cnt[o] > n_children)
      return fail(e, TOHC_E_INVALID, "upload_scene: object %d has a bad child range", o);
    if (!obj_compound[o] && obj_child_cnt[o] != 1) return fail(e, TOHC_E_INVALID, "upload_scene: bare convex object %d must have one child", o);
    for (int k = 0; k < obj_child_cnt[o]; ++k) {
      e->h_child_obj[obj_child_off[o] + k] = o;
      e->h_child_idx[obj_child_off[o] + k] = k;
    }
  }
  for (int c = 0; c < n_children; ++c)
    if (e->h_child_obj[c] < 0) return fail(e, TOHC_E_INVALID, "upload_scene: child %d belongs to no object", c);
  e->h_allowed.assign((size_t)n_objects * n_objects, 1);
  // "poses" mode: every object moves
  e->h_obj_dynP.resize(n_objects); std::iota(e->h_obj_dynP.begin(), e->h_obj_dynP.end(), 0);
  e->h_child_dynP.resize(n_children); std::iota(e->h_child_dynP.begin(), e->h_child_dynP.end(), 0);
  e->h_dyn_childrenP = e->h_child_dynP;
  hipStream_t s = e->stream;
  HIPCHK(upload(e->d_shape_kind, e->h_shape_kind, s)); HIPCHK(upload(e->d_shape_voff, e->h_shape_voff, s)); HIPCHK(upload(e->d_shape_vcnt, e->h_shape_vcnt, s));
  HIPCHK(upload(e->d_shape_dims, e->h_shape_dims, s)); HIPCHK(upload(e->d_shape_lmin, e->h_shape_lmin, s)); HIPCHK(upload(e->d_shape_lmax, e->h_shape_lmax, s));
  {
    // SoA vertex arrays, padded so that a 64-lane load past the last hull stays in bounds
    std::vector<float> px((size_t)n_verts + 64, 0.f), py((size_t)n_verts + 64, 0.f), pz((size_t)n_verts + 64, 0.f);
    for (int i = 0; i < n_verts; ++i) { px[i] = vx[i]; py[i] = vy[i]; pz[i] = vz[i]; }
    HIPCHK(upload(e->d_vx, px, s)); HIPCHK(upload(e->d_vy, py, s)); HIPCHK(upload(e->d_vz, pz, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(upload(e->d_child_shape, e->h_child_shape, s)); HIPCHK(upload(e->d_child_obj, e->h_child_obj, s)); HIPCHK(upload(e->d_child_idx, e->h_child_idx, s));
  HIPCHK(upload(e->d_child_localT, e->h_child_localT, s));
  HIPCHK(upload(e->d_obj_off, e->h_obj_off, s)); HIPCHK(upload(e->d_obj_cnt, e->h_obj_cnt, s)); HIPCHK(upload(e->d_obj_compound, e->h_obj_compound, s));
  HIPCHK(upload(e->d_obj_group, e->h_obj_group, s)); HIPCHK(upload(e->d_allowed, e->h_allowed, s));
  HIPCHK(upload(e->d_obj_dynP, e->h_obj_dynP, s)); HIPCHK(upload(e->d_child_dynP, e->h_child_dynP, s)); HIPCHK(upload(e->d_dyn_childrenP, e->h_dyn_childrenP, s));
  HIPCHK(hipStreamSynchronize(s));
  e->have_scene = true;
  e->have_robot = false;
  return refreshStatic(e);
}

int tohc_upload_robot(tohc_engine *e, int n_links, const int32_t *parent, const int32_t *jtype, const double *T_origin, const double *axis,
                      const int32_t *dof, const double *mult, const double *offs, const int32_t *link_obj, int n_dof, const double *q_robot,
                      const double *base_T, int n_joint_inds, const int32_t *joint_inds, int affine, const double *rot_axis) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload_robot: upload the scene first");
  if (n_links <= 0 || !parent || !jtype || !T_origin || !axis || !dof || !mult || !offs || !link_obj || !base_T || n_joint_inds < 0 ||
      (n_joint_inds > 0 && !joint_inds) || (n_dof > 0 && !q_robot))
    return fail(e, TOHC_E_INVALID, "upload_robot: null or empty input");
  if (affine & ~15) return fail(e, TOHC_E_UNSUPPORTED, "upload_robot: only DOF_X|DOF_Y|DOF_Z|DOF_RotationAxis affine DOFs are supported");
  HIPCHK(hipSetDevice(e->device));
  e->n_links = n_links; e->n_joint_inds = n_joint_inds; e->affine = affine; e->n_dof = n_dof;
  e->K = n_joint_inds + ((affine & 1) != 0) + ((affine & 2) != 0) + ((affine & 4) != 0) + ((affine & 8) != 0);
  if (e->K <= 0) return fail(e, TOHC_E_INVALID, "upload_robot: no active degrees of freedom");
  e->h_parent.assign(parent, parent + n_links); e->h_jtype.assign(jtype, jtype + n_links); e->h_dof.assign(dof, dof + n_links);
  e->h_link_obj.assign(link_obj, link_obj + n_links);
  e->h_dof_col.assign(n_links, -1);
  for (int i = 0; i < n_links; ++i) {
    if (parent[i] >= i) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: parents must precede children", i);
    if (dof[i] >= n_dof) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: dof %d out of range", i, dof[i]);
    if (link_obj[i] >= e->n_objects) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: object %d out of range", i, link_obj[i]);
    if (jtype[i] != TOHC_J_FIXED && dof[i] >= 0)
      for (int a = 0; a < n_joint_inds; ++a)
        if (joint_inds[a] == dof[i]) { e->h_dof_col[i] = a; break; }
  }
  memcpy(e->base_T, base_T, sizeof e->base_T);
  e->rot_axis[0] = rot_axis ? rot_axis[0] : 0; e->rot_axis[1] = rot_axis ? rot_axis[1] : 0; e->rot_axis[2] = rot_axis ? rot_axis[2] : 1;
  // dynamic map (robot mode) and query children = GetAffectedLinks(only_with_geom) (configuration_space.cpp:110-118)
  e->h_obj_dynR.assign(e->n_objects, -1); e->h_child_dynR.assign(e->n_children, -1);
  e->h_dyn_childrenR.clear(); e->h_qchildrenR.clear();
  e->h_link_dyn.assign(n_links, -1); e->h_obj_jaclink.assign(e->n_objects, -1);
  int nd = 0;
  for (int i = 0; i < n_links; ++i) {
    int o = link_obj[i];
    if (o < 0) continue;
    if (e->h_obj_dynR[o] >= 0) return fail(e, TOHC_E_INVALID, "upload_robot: object %d is attached to two links", o);
    e->h_obj_dynR[o] = nd; e->h_link_dyn[i] = nd; ++nd;
    bool affected = affine > 0;
    for (int j = i; j >= 0 && !affected; j = parent[j])
      if (e->h_dof_col[j] >= 0) affected = true;
    for (int k = 0; k < e->h_obj_cnt[o]; ++k) {
      int c = e->h_obj_off[o] + k;
      e->h_child_dynR[c] = (int)e->h_dyn_childrenR.size();
      e->h_dyn_childrenR.push_back(c);
      if (affected) e->h_qchildrenR.push_back(c);
    }
    if (affected) e->h_obj_jaclink[o] = i;
  }
  e->n_dyn_objR = nd; e->n_dyn_childrenR = (int)e->h_dyn_childrenR.size();
  hipStream_t s = e->stream;
  HIPCHK(upload(e->d_parent, e->h_parent, s)); HIPCHK(upload(e->d_jtype, e->h_jtype, s)); HIPCHK(upload(e->d_dof, e->h_dof, s));
  HIPCHK(upload(e->d_dof_col, e->h_dof_col, s)); HIPCHK(upload(e->d_link_dyn, e->h_link_dyn, s)); HIPCHK(upload(e->d_obj_jaclink, e->h_obj_jaclink, s));
  HIPCHK(upload(e->d_T_origin, T_origin, (size_t)7 * n_links, s)); HIPCHK(upload(e->d_axis, axis, (size_t)3 * n_links, s));
  HIPCHK(upload(e->d_mult, mult, (size_t)n_links, s)); HIPCHK(upload(e->d_offs, offs, (size_t)n_links, s));
  {
    std::vector<double> q(std::max(n_dof, 1), 0.0);
    for (int i = 0; i < n_dof; ++i) q[i] = q_robot[i];
    HIPCHK(upload(e->d_q_robot, q, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(upload(e->d_obj_dynR, e->h_obj_dynR, s)); HIPCHK(upload(e->d_child_dynR, e->h_child_dynR, s));
  HIPCHK(upload(e->d_dyn_childrenR, e->h_dyn_childrenR, s)); HIPCHK(upload(e->d_qchildrenR, e->h_qchildrenR, s));
  HIPCHK(hipStreamSynchronize(s));
  e->have_robot = true;
  return TOHC_OK;
}

int tohc_set_allowed_matrix(tohc_engine *e, int n_objects, const uint8_t *allowed) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "set_allowed_matrix: upload the scene first");
  if (n_objects != e->n_objects || !allowed) return fail(e, TOHC_E_INVALID, "set_allowed_matrix: expected a %d x %d matrix", e->n_objects, e->n_objects);
  HIPCHK(hipSetDevice(e->device));
  e->h_allowed.assign(allowed, allowed + (size_t)n_objects * n_objects);
  HIPCHK(upload(e->d_allowed, e->h_allowed, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return TOHC_OK;
}

int tohc_set_contact_distance(tohc_engine *e, float d) {
  if (!e) return TOHC_E_INVALID;
  if (!(d >= 0)) return fail(e, TOHC_E_INVALID, "set_contact_distance: negative or NaN distance");
  e->d = d;
  e->breaking = (float)(2.001 * (double)d);  // gContactBreakingThreshold, bullet_collision_checker.cpp:453
  if (e->have_scene) {
    HIPCHK(hipSetDevice(e->device));
    return refreshStatic(e);
  }
  return TOHC_OK;
}
double tohc_get_contact_distance(const tohc_engine *e) { return e ? (double)e->d : 0.0; }

int tohc_set_capacity(tohc_engine *e, int64_t max_pairs, int64_t max_records) {
  if (!e || max_pairs < 1 || max_records < 1 || max_pairs > ((int64_t)1 << 30) || max_records > ((int64_t)1 << 30)) return TOHC_E_INVALID;
  e->max_pairs = (size_t)max_pairs;
  e->max_records = (size_t)max_records;
  return TOHC_OK;
}

int tohc_get_counters(const tohc_engine *e, tohc_counters *out) {
  if (!e || !out) return TOHC_E_INVALID;
  *out = e->counters;
  return TOHC_OK;
}

static DevScene devScene(tohc_engine *e) {
  DevScene s;
  s.n_shapes = e->n_shapes; s.n_children = e->n_children; s.n_objects = e->n_objects;
  s.shape_kind = e->d_shape_kind.p; s.shape_voff = e->d_shape_voff.p; s.shape_vcnt = e->d_shape_vcnt.p;
  s.shape_dims = e->d_shape_dims.p; s.shape_lmin = e->d_shape_lmin.p; s.shape_lmax = e->d_shape_lmax.p;
  s.vx = e->d_vx.p; s.vy = e->d_vy.p; s.vz = e->d_vz.p;
  s.child_shape = e->d_child_shape.p; s.child_obj = e->d_child_obj.p; s.child_idx = e->d_child_idx.p; s.child_localT = e->d_child_localT.p;
  s.obj_off = e->d_obj_off.p; s.obj_cnt = e->d_obj_cnt.p; s.obj_compound = e->d_obj_compound.p; s.obj_group = e->d_obj_group.p;
  s.allowed = e->d_allowed.p; s.stat_xf = e->d_stat_xf.p; s.stat_aabb = e->d_stat_aabb.p;
  return s;
}
static DevRobot devRobot(tohc_engine *e) {
  DevRobot r;
  memset(&r, 0, sizeof r);
  r.n_links = e->n_links; r.K = e->K; r.n_joint_inds = e->n_joint_inds; r.affine = e->affine;
  r.parent = e->d_parent.p; r.jtype = e->d_jtype.p; r.dof = e->d_dof.p; r.dof_col = e->d_dof_col.p; r.link_dyn = e->d_link_dyn.p;
  r.T_origin = e->d_T_origin.p; r.axis = e->d_axis.p; r.mult = e->d_mult.p; r.offs = e->d_offs.p; r.q_robot = e->d_q_robot.p;
  r.obj_jaclink = e->d_obj_jaclink.p;
  memcpy(r.base_T, e->base_T, sizeof r.base_T);
  memcpy(r.rot_axis, e->rot_axis, sizeof r.rot_axis);
  return r;
}

// -----------------------------------------------------------------------------------------------------------------
// stage -> enqueue -> sync -> fetch.  The one-shot batch calls are exactly these four in a row.
// -----------------------------------------------------------------------------------------------------------------
static int syncEngine(tohc_engine *e) {
  hipStream_t s = e->stream;
  if (e->pending.empty()) { HIPCHK(hipStreamSynchronize(s)); return TOHC_OK; }
  HIPCHK(hipMemcpyAsync(e->h_cnt.p, e->d_cnt.p, sizeof(Counters) * RING, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(e->h_nrec.p, e->d_nrec.p, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  bool overflow = false;
  unsigned int worst_pairs = 0;
  tohc_counters &c = e->counters;
  c.aabb_tests = c.pair_evals = c.epa_evals = 0;
  c.ms_last_call = 0;
  for (const auto &pd : e->pending) {
    const Counters &k = e->h_cnt.p[pd.slot];
    overflow |= k.overflow != 0 || k.n_pairs > e->max_pairs;
    worst_pairs = std::max(worst_pairs, k.n_pairs);
    c.aabb_tests += pd.aabb_tests; c.pair_evals += k.n_pairs; c.epa_evals += k.n_epa;
    c.cum_aabb_tests += pd.aabb_tests; c.cum_pair_evals += k.n_pairs; c.cum_epa_evals += k.n_epa;
    float ms[5] = {0, 0, 0, 0, 0};
    hipEvent_t *ev = &e->events[(size_t)pd.slot * 6];
    for (int i = 0; i < 5; ++i) HIPCHK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    c.cum_ms_fk += ms[0]; c.cum_ms_children += ms[1]; c.cum_ms_broadphase += ms[2]; c.cum_ms_gjk += ms[3]; c.cum_ms_epa += ms[4];
    c.ms_last_call += ms[0] + ms[1] + ms[2] + ms[3] + ms[4];
    c.cum_launches += 1;
  }
  e->pending.clear();
  const unsigned int nrec = *e->h_nrec.p;
  c.contacts = nrec;
  if (overflow || nrec > e->max_records)
    return fail(e, TOHC_E_CAPACITY, "capacity exceeded: %u pairs in one enqueue (max %zu), %u records (max %zu); raise with tohc_set_capacity",
                worst_pairs, e->max_pairs, nrec, e->max_records);
  return TOHC_OK;
}

static int64_t countAabbTests(const tohc_engine *e, const std::vector<int> &qchildren, const CallParams &cp) {
  // tuples the broadphase examines = units x (query child, other child) combinations that pass the object filters
  int64_t per = 0;
  for (int q : qchildren) {
    int Q = e->h_child_obj[q];
    for (int X = 0; X < e->n_objects; ++X) {
      if (X == Q) continue;
      int g = e->h_obj_group[X];
      if (cp.cast ? !(g & TOHC_GROUP_KINBODY) : !(g & (short)cp.filter_mask)) continue;
      if (!e->h_allowed[(size_t)Q * e->n_objects + X]) continue;
      per += e->h_obj_cnt[X];
    }
  }
  return per * cp.n_cast_sets;
}

// broadphase -> GJK -> EPA for one enqueue; `ev` = this enqueue's events [2..5] are recorded here
static int launchNarrow(tohc_engine *e, const DynMap &dm, const int *d_qchildren, const std::vector<int> &h_qchildren, CallParams cp,
                        bool robotMode, int slot, bool append) {
  const int nQ = (int)h_qchildren.size();
  const int K = robotMode ? e->K : 1;
  hipStream_t s = e->stream;
  hipEvent_t *ev = &e->events[(size_t)slot * 6];
  HIPCHK(hipMemsetAsync(e->d_cnt.p + slot, 0, sizeof(Counters), s));
  if (!append) HIPCHK(hipMemsetAsync(e->d_nrec.p, 0, sizeof(unsigned int), s));
  cp.max_pairs = (unsigned int)e->max_pairs; cp.max_records = (unsigned int)e->max_records;
  HIPCHK(e->d_pairs.reserve(e->max_pairs)); HIPCHK(e->d_epaq.reserve(e->max_pairs));
  HIPCHK(e->d_records.reserve(e->max_records));
  HIPCHK(e->d_grads.reserve(e->max_records * 2 * (size_t)K));
  HIPCHK(hipEventRecord(ev[2], s));
  if (nQ > 0 && cp.n_cast_sets > 0) {
    DevScene sc = devScene(e);
    const int BT = 256;
    const int xTiles = (e->n_children + BT - 1) / BT;
    const size_t blocks = (size_t)cp.n_cast_sets * nQ * xTiles;
    if (blocks > 0x7fffffffull) return fail(e, TOHC_E_CAPACITY, "broadphase grid too large (%zu blocks)", blocks);
    const bool sep1 = cp.gap == 0 && cp.cast;  // poses mode keeps the end-of-sweep poses in their own tables
    hipLaunchKernelGGL(k_broadphase, dim3((unsigned)blocks), dim3(BT), 0, s, sc, dm, cp, d_qchildren, nQ, e->d_dynAabb.p,
                       sep1 ? e->d_dynAabb1.p : e->d_dynAabb.p, e->d_pairs.p, e->d_cnt.p + slot);
    HIPCHK(hipEventRecord(ev[3], s));
    NarrowArgs a;
    a.sc = sc; a.dm = dm; a.rb = devRobot(e); a.cp = cp;
    a.dynXf0 = e->d_dynXf.p; a.dynXf1 = sep1 ? e->d_dynXf1.p : e->d_dynXf.p;
    a.linkQT = robotMode ? e->d_linkQT.p : nullptr;
    a.dofs = robotMode ? e->d_dofs.p : nullptr;
    a.pairs = e->d_pairs.p; a.epaq = e->d_epaq.p; a.cnt = e->d_cnt.p + slot; a.n_records = e->d_nrec.p;
    a.records = e->d_records.p; a.grads = e->d_grads.p;
    const int gridN = 256 * 8;
    if (e->group == 64) hipLaunchKernelGGL((k_gjk<64>), dim3(gridN), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gjk<16>), dim3(gridN), dim3(256), 0, s, a);
    HIPCHK(hipEventRecord(ev[4], s));
    if (e->group == 64) hipLaunchKernelGGL((k_epa<64, 64>), dim3(gridN), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_epa<16, 64>), dim3(gridN), dim3(64), 0, s, a);
    HIPCHK(hipEventRecord(ev[5], s));
    HIPCHK(hipGetLastError());
  } else {
    HIPCHK(hipEventRecord(ev[3], s)); HIPCHK(hipEventRecord(ev[4], s)); HIPCHK(hipEventRecord(ev[5], s));
  }
  tohc_engine::Pending pd;
  pd.slot = slot;
  pd.aabb_tests = countAabbTests(e, h_qchildren, cp);
  e->pending.push_back(pd);
  e->last_robot_mode = robotMode;
  e->last_want_jac = robotMode && cp.want_jac;
  return TOHC_OK;
}

static int nextSlot(tohc_engine *e, int *slot) {
  if ((int)e->pending.size() >= RING) {
    int rc = syncEngine(e);
    if (rc != TOHC_OK) return rc;
  }
  *slot = (int)e->pending.size();
  return TOHC_OK;
}

int tohc_stage_dofs(tohc_engine *e, const double *dofs, int B, int T) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene || !e->have_robot) return fail(e, TOHC_E_STATE, "upload the scene and the robot first");
  if (!dofs || B < 1 || T < 1) return fail(e, TOHC_E_INVALID, "dofs is null or B/T < 1");
  HIPCHK(hipSetDevice(e->device));
  const size_t n_sets = (size_t)B * T;
  HIPCHK(upload(e->d_dofs, dofs, n_sets * e->K, e->stream));
  HIPCHK(e->d_linkQT.reserve(n_sets * e->n_links * 7));
  HIPCHK(e->d_objQT.reserve(n_sets * std::max(e->n_dyn_objR, 1) * 7));
  HIPCHK(e->d_dynXf.reserve(n_sets * std::max(e->n_dyn_childrenR, 1) * 12));
  HIPCHK(e->d_dynAabb.reserve(n_sets * std::max(e->n_dyn_childrenR, 1) * 6));
  HIPCHK(hipStreamSynchronize(e->stream));  // the caller's buffer may go away
  e->staged_B = B; e->staged_T = T; e->staged = true;
  return TOHC_OK;
}

static int enqueueRobot(tohc_engine *e, int gap, int cast, int filter_mask, int want_jac, int append) {
  if (!e) return TOHC_E_INVALID;
  if (!e->staged) return fail(e, TOHC_E_STATE, "stage the DOF vectors first (tohc_stage_dofs)");
  if (cast && gap < 1) return fail(e, TOHC_E_INVALID, "cast: gap must be >= 1");
  HIPCHK(hipSetDevice(e->device));
  int slot;
  int rc = nextSlot(e, &slot);
  if (rc != TOHC_OK) return rc;
  const int B = e->staged_B, T = e->staged_T;
  const size_t n_sets = (size_t)B * T;
  hipStream_t s = e->stream;
  hipEvent_t *ev = &e->events[(size_t)slot * 6];
  HIPCHK(hipEventRecord(ev[0], s));
  hipLaunchKernelGGL(k_fk, dim3((unsigned)((n_sets + 63) / 64)), dim3(64), 0, s, devRobot(e), e->d_dofs.p, (int)n_sets, e->d_linkQT.p, e->d_objQT.p,
                     e->n_dyn_objR);
  HIPCHK(hipEventRecord(ev[1], s));
  DynMap dm;
  dm.obj_dyn = e->d_obj_dynR.p; dm.child_dyn = e->d_child_dynR.p; dm.n_dyn_obj = e->n_dyn_objR; dm.n_dyn_children = e->n_dyn_childrenR;
  const size_t tot = n_sets * e->n_dyn_childrenR;
  if (tot > 0)
    hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_dyn_childrenR.p, e->d_objQT.p,
                       (int)n_sets, e->d, e->d_dynXf.p, e->d_dynAabb.p);
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = T; cp.gap = cast ? gap : 0; cp.cast = cast; cp.filter_mask = filter_mask; cp.want_jac = want_jac; cp.d = e->d; cp.breaking = e->breaking;
  cp.n_cast_sets = cast ? B * std::max(T - gap, 0) : B * T;
  return launchNarrow(e, dm, e->d_qchildrenR.p, e->h_qchildrenR, cp, true, slot, append != 0);
}

int tohc_enqueue_links_vs_all(tohc_engine *e, int filter_mask, int want_jacobians, int append) {
  return enqueueRobot(e, 0, 0, filter_mask, want_jacobians, append);
}
int tohc_enqueue_cast_vs_all(tohc_engine *e, int gap, int want_jacobians, int append) {
  return enqueueRobot(e, gap, 1, -1, want_jacobians, append);
}

int tohc_sync(tohc_engine *e) {
  if (!e) return TOHC_E_INVALID;
  HIPCHK(hipSetDevice(e->device));
  return syncEngine(e);
}

int tohc_fetch(tohc_engine *e, const tohc_record **records, const double **grads, int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (records) *records = nullptr;
  if (grads) *grads = nullptr;
  HIPCHK(hipSetDevice(e->device));
  int rc = syncEngine(e);
  if (rc != TOHC_OK) return rc;
  const size_t n = *e->h_nrec.p;
  if (n == 0) return TOHC_OK;
  hipStream_t s = e->stream;
  const int K = e->last_robot_mode ? e->K : 1;
  HIPCHK(e->h_records.reserve(n)); HIPCHK(e->h_sorted.reserve(n));
  HIPCHK(hipMemcpyAsync(e->h_records.p, e->d_records.p, n * sizeof(tohc_record), hipMemcpyDeviceToHost, s));
  const bool haveG = e->last_want_jac;
  if (haveG) {
    HIPCHK(e->h_grads.reserve(n * 2 * K)); HIPCHK(e->h_gsorted.reserve(n * 2 * K));
    HIPCHK(hipMemcpyAsync(e->h_grads.p, e->d_grads.p, n * 2 * K * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  // canonical order: Bullet's own order follows DBVT traversal and is history dependent.
  // discrete records (cctype none) sort before cast records of the same (problem, t) when both were appended.
  std::vector<uint32_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0u);
  const tohc_record *r = e->h_records.p;
  std::sort(idx.begin(), idx.end(), [r](uint32_t i, uint32_t j) {
    const tohc_record &a = r[i], &b = r[j];
    if (a.problem != b.problem) return a.problem < b.problem;
    if (a.t != b.t) return a.t < b.t;
    const int ca = a.cctype != 0, cb = b.cctype != 0;
    if (ca != cb) return ca < cb;
    if (a.idA != b.idA) return a.idA < b.idA;
    if (a.childA != b.childA) return a.childA < b.childA;
    if (a.idB != b.idB) return a.idB < b.idB;
    return a.childB < b.childB;
  });
  for (size_t i = 0; i < n; ++i) {
    e->h_sorted.p[i] = r[idx[i]];
    if (haveG) memcpy(e->h_gsorted.p + i * 2 * K, e->h_grads.p + (size_t)idx[i] * 2 * K, sizeof(double) * 2 * K);
  }
  *n_records = (int)n;
  if (records) *records = e->h_sorted.p;
  if (grads) *grads = haveG ? e->h_gsorted.p : nullptr;
  return TOHC_OK;
}

int tohc_links_vs_all_batch(tohc_engine *e, const double *dofs, int B, int T, int filter_mask, int want_jacobians, const tohc_record **records,
                            const double **grads, int *n_records) {
  int rc = tohc_stage_dofs(e, dofs, B, T);
  if (rc == TOHC_OK) rc = tohc_enqueue_links_vs_all(e, filter_mask, want_jacobians, 0);
  if (rc == TOHC_OK) rc = tohc_fetch(e, records, grads, n_records);
  return rc;
}
int tohc_cast_vs_all_batch(tohc_engine *e, const double *dofs, int B, int T, int gap, int want_jacobians, const tohc_record **records,
                           const double **grads, int *n_records) {
  int rc = tohc_stage_dofs(e, dofs, B, T);
  if (rc == TOHC_OK) rc = tohc_enqueue_cast_vs_all(e, gap, want_jacobians, 0);
  if (rc == TOHC_OK) rc = tohc_fetch(e, records, grads, n_records);
  return rc;
}

static int posesQuery(tohc_engine *e, const double *obj_T, const double *cast_T1, int n_sets, const int32_t *query, int n_query, int filter_mask,
                      int cast, const tohc_record **records, int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  if (!obj_T || n_sets < 1 || n_query < 0 || (n_query > 0 && !query) || (cast && !cast_T1)) return fail(e, TOHC_E_INVALID, "null pose or query input");
  HIPCHK(hipSetDevice(e->device));
  std::vector<int> qch;
  std::vector<char> seen(e->n_objects, 0);
  for (int i = 0; i < n_query; ++i) {
    int o = query[i];
    if (o < 0 || o >= e->n_objects) return fail(e, TOHC_E_INVALID, "query object %d out of range", o);
    if (cast && seen[o]) return fail(e, TOHC_E_INVALID, "cast: query object %d listed twice", o);
    seen[o] = 1;
    for (int k = 0; k < e->h_obj_cnt[o]; ++k) qch.push_back(e->h_obj_off[o] + k);
  }
  int slot;
  int rc = nextSlot(e, &slot);
  if (rc != TOHC_OK) return rc;
  e->staged = false;  // the pose tables below overwrite the staged robot tables
  hipStream_t s = e->stream;
  hipEvent_t *ev = &e->events[(size_t)slot * 6];
  HIPCHK(hipEventRecord(ev[0], s));
  HIPCHK(hipEventRecord(ev[1], s));
  const size_t nobj = e->n_objects, nch = e->n_children;
  HIPCHK(upload(e->d_objQT, obj_T, (size_t)n_sets * nobj * 7, s));
  HIPCHK(upload(e->d_qchildrenP, qch, s));
  HIPCHK(e->d_dynXf.reserve((size_t)n_sets * nch * 12)); HIPCHK(e->d_dynAabb.reserve((size_t)n_sets * nch * 6));
  DynMap dm;
  dm.obj_dyn = e->d_obj_dynP.p; dm.child_dyn = e->d_child_dynP.p; dm.n_dyn_obj = (int)nobj; dm.n_dyn_children = (int)nch;
  const size_t tot = (size_t)n_sets * nch;
  hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_dyn_childrenP.p, e->d_objQT.p, n_sets,
                     e->d, e->d_dynXf.p, e->d_dynAabb.p);
  std::vector<double> T1;
  if (cast) {
    // end-of-sweep poses: the query links move, everything else keeps its pose
    T1.assign(obj_T, obj_T + (size_t)n_sets * nobj * 7);
    for (int st = 0; st < n_sets; ++st)
      for (int i = 0; i < n_query; ++i) memcpy(&T1[((size_t)st * nobj + query[i]) * 7], cast_T1 + ((size_t)st * n_query + i) * 7, 7 * sizeof(double));
    HIPCHK(upload(e->d_objQT1, T1, s));
    HIPCHK(e->d_dynXf1.reserve((size_t)n_sets * nch * 12)); HIPCHK(e->d_dynAabb1.reserve((size_t)n_sets * nch * 6));
    hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_dyn_childrenP.p, e->d_objQT1.p,
                       n_sets, e->d, e->d_dynXf1.p, e->d_dynAabb1.p);
  }
  HIPCHK(hipStreamSynchronize(s));  // caller-owned / local host buffers may go away
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = 1; cp.gap = 0; cp.cast = cast; cp.filter_mask = filter_mask; cp.want_jac = 0; cp.d = e->d; cp.breaking = e->breaking;
  cp.n_cast_sets = n_sets;
  rc = launchNarrow(e, dm, e->d_qchildrenP.p, qch, cp, false, slot, false);
  if (rc != TOHC_OK) return rc;
  const double *g = nullptr;
  return tohc_fetch(e, records, &g, n_records);
}

int tohc_links_vs_all_poses(tohc_engine *e, const double *obj_T, int n_sets, const int32_t *query, int n_query, int filter_mask,
                            const tohc_record **records, int *n_records) {
  return posesQuery(e, obj_T, nullptr, n_sets, query, n_query, filter_mask, 0, records, n_records);
}
int tohc_cast_vs_all_poses(tohc_engine *e, const double *obj_T, const double *cast_T1, int n_sets, const int32_t *query, int n_query,
                           const tohc_record **records, int *n_records) {
  return posesQuery(e, obj_T, cast_T1, n_sets, query, n_query, -1, 1, records, n_records);
}

int tohc_forward_kinematics(tohc_engine *e, const double *dofs, int n_cfg, double *link_T) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_robot) return fail(e, TOHC_E_STATE, "upload the robot first");
  if (!dofs || !link_T || n_cfg < 1) return fail(e, TOHC_E_INVALID, "null input");
  HIPCHK(hipSetDevice(e->device));
  int rc = syncEngine(e);
  if (rc != TOHC_OK) return rc;
  e->staged = false;
  hipStream_t s = e->stream;
  HIPCHK(upload(e->d_dofs, dofs, (size_t)n_cfg * e->K, s));
  HIPCHK(e->d_linkQT.reserve((size_t)n_cfg * e->n_links * 7));
  hipLaunchKernelGGL(k_fk, dim3((unsigned)((n_cfg + 63) / 64)), dim3(64), 0, s, devRobot(e), e->d_dofs.p, n_cfg, e->d_linkQT.p, (double *)nullptr, 0);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(link_T, e->d_linkQT.p, (size_t)n_cfg * e->n_links * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return TOHC_OK;
}

}  // extern "C"
