// tohc_kernels.h -- device-side data views and the gfx950 kernels of the collision-linearization engine.
//
//   k_fk            one 64-thread block per configuration: joint motions in parallel, then every link composes its own
//                   chain root-first (same association as a serial walk, so results equal the serial oracle bit for
//                   bit); the same thread emits the fp32 pose + d-inflated cull box of the link's geometry
//   k_dyn_children  "poses" mode only: object poses given by the caller -> child poses + cull boxes
//   k_broadphase    one block per (unit, query child): streams the SoA box table of the static children (coalesced),
//                   queues hits in LDS and reserves space in the pair list with ONE atomic per flush, on one of
//                   NSHARD counters (a single word saturates at ~88 atomics/us, MI355X_MICROARCH.md "dequeue")
//   k_gjk           one group of G lanes per pair: GJK closest points (tohc_narrow.h); separated pairs that pass the
//                   contact filter go to the hit list, penetrating / degenerate ones to the EPA queue
//   k_tail_pen      the queued pairs of EVERY open pass (TailArgs), one lane per pair: GJK2 of the penetration solver;
//                   origin outside -> Distance fall-back, done; origin inside -> on to k_tail_epa with the GJK2 state
//   k_tail_epa      EPA proper, one group of 16 lanes per pair, face / vertex pools in LDS -> hit list
//   k_tail_range    reserves each open pass's record range in the shared slab, in enqueue order
//   k_finish        one group per hit: cast classification, record write, fused Jacobian rows (fp64); the p-th hit
//                   becomes record base+p, so no atomics here
// All queues are sharded NSHARD ways by producer block; consumers walk them through a 16-entry prefix.
#pragma once
#include "../../include/trajopt_hipcc.h"
#include "tohc_narrow.h"

namespace tohc {

#define TOHC_NSHARD 16

struct ChildDesc {   // 64 B = one cache line per child: cuts the dependent-load chain of the pair set-up to
  int obj, idx;      // pair -> desc -> vertices
  int kind, compound;
  int voff, vcnt;
  int dynR, group;   // slot among the robot's moving children (robot mode) or -1; the object's filter group
  float dims[3];
  int voff8;         // a hull's first block in DevScene::v8
  float sph[4];      // bounding sphere in the child's frame: centre of its local box, radius (slightly inflated)
};
struct CullSph {     // 32 B per child: all the sphere stage of the pre-test needs of a child that never moves
  float c[3], r;     // its bounding sphere in WORLD coordinates (static pose), radius as ChildDesc::sph[3]
  int obj, group;
  int pad[2];
};
struct DevScene {
  int n_shapes, n_children, n_objects;
  const int *shape_kind, *shape_voff, *shape_vcnt;
  const float *shape_dims, *shape_lmin, *shape_lmax;  // [n][3]
  const float *vx, *vy, *vz;
  const F4 *v4;                                       // the same vertices as 16-byte records (groups of 4 / 8 lanes)
  const float *v8;                                    // ... and in blocks of 8 (tohc_narrow.h, Cvx<1>): the lane-per-pair kernels' table
  const int *child_shape, *child_obj, *child_idx;
  const float *child_localT;                          // [n][12] rows of the 3x3 then origin
  const int *obj_off, *obj_cnt, *obj_compound, *obj_group;
  const unsigned char *allowed;
  const struct ChildDesc *child_desc;                 // [n_children] everything the narrowphase needs about a child, packed
  const float *stat_xf;                               // [n_children][12] world transform when the object is static
  const float *stat_aabb;                             // [n_children][6]
  const struct CullSph *stat_sph;                     // [n_children] world bounding sphere + filter words, as if static
};
struct DynMap {          // which objects / children move per set (robot links, or everything in "poses" mode)
  const int *obj_dyn;    // [n_objects]  -> slot or -1
  const int *child_dyn;  // [n_children] -> slot or -1
  int n_dyn_obj, n_dyn_children;
  const int *dyn_children;   // [n_dyn_children] child ids
  const int *stat_children;  // [n_stat] child ids that never move in this mode
  const float *stat_soa;     // [6][n_stat_pad] their cull boxes, SoA, Morton order, padded to whole clusters of 64
  int n_stat, n_stat_pad;
  const float *cl_soa;       // [6][n_cl_pad] box of each cluster of 64 consecutive table entries
  int n_cl, n_cl_pad;
};
#define TOHC_FAST_JAC_MAXK 64
// k_finish's tables, flattened so that a contact reaches its joints' poses in two loads after its pair record:
// child -> (link whose chain gives the Jacobian, its joints' range) and, per link, the joints of its chain that move an
// optimisation column (fixed joints and joints outside the problem's columns left out)
struct JacJoint { int link, col, revolute, pad; double mult, pad2; };
struct DevRobot {
  int n_links, K, n_joint_inds, affine;
  int fast_jac;   // 1: no chain has two joints on one column and K <= TOHC_FAST_JAC_MAXK (k_finish's lane-per-joint epilogue applies)
  const int *parent, *jtype, *dof, *dof_col, *link_dyn, *link_obj;
  const int *chain_off, *chain;                          // ancestors of each link, root first, the link itself last
  const double *T_origin, *axis, *mult, *offs, *q_robot;
  const int *obj_jaclink;                                // [n_objects] link whose chain gives the Jacobian, or -1
  const int4 *child_jac;                                 // [n_children] (that link, first, one past the last of its JacJoint records, 0)
  const JacJoint *jac_joint;
  double base_T[7];
  double rot_axis[3];
};
struct PairRec { int set, q, x, pad; };
struct EpaRec { int pair; int isValid; float distance; float n[3]; float pB[3]; };
struct HitRec { int pair; float distance; float n[3]; float pB[3]; };
struct PenRec { EpaRec e; PenHand h; };   // k_pen -> k_epa
// a GJK run after a few passes of its loop, everything the next pass reads (k_gjk -> k_gjk_cont): the simplex holds at most
// three vertices then (a fourth ends the loop), its closest points are up to date
struct GjkCont {
  unsigned int pos;   // the pair's place in the pair list
  int curIter, n, cachedValid;
  float squaredDistance;
  float sep[3], lastW[3], cP1[3], cP2[3];
  float W[3][3], P[3][3], Q[3][3];
};
#define TOHC_CONT_DIV 8                    // the k_gjk -> k_gjk_cont queue holds 1/8 of the pair capacity (a full queue: the run stays where it is)
#ifndef TOHC_EPA_WAVES
#define TOHC_EPA_WAVES 2
#endif
//                                         k_epa is bound by the latency of dependent instructions: more resident waves, more pairs in flight
#define TOHC_PEN_DIV 8                     // the k_pen -> k_epa queue holds 1/8 of the pair capacity
// One per enqueue.  Every shard of the queue counters sits on its own 128-byte line: atomics on words of ONE line
// serialise (~88 per microsecond, MI355X_MICROARCH "dequeue"), so 16 shards inside one line were no shards at all --
// the per-wavefront adds of k_broadphase's discrete pass alone cost ~200 us that way.
struct CounterShard {
  unsigned int n_pairs, n_epa, n_hits;
  unsigned int n_pen;    // pairs k_pen hands to k_epa (origin inside the Minkowski difference)
  unsigned int n_cand;   // box-test survivors = pair evaluations; n_pairs = those the pre-test passes on to k_gjk
  // work done, for the flop side of the roofline (flops per pair-eval = iterations x (5 (nA + nB) + 200)):
  unsigned int n_gjk_steps;   // passes of the GJK loop body in k_gjk (one Minkowski support query each)
  unsigned int n_pen_sups;    // Minkowski support queries of GJK2 / the Distance fall-back in k_pen
  unsigned int n_epa_sups;    // the same in k_epa (one per EPA iteration, plus EncloseOrigin's)
  unsigned int n_epa_redone;  // EPA runs repeated on the full pools
  unsigned int n_exact;       // pairs k_broadphase's sphere stage leaves for k_exact (NarrowArgs.cands)
  unsigned int n_cont;        // GJK runs k_gjk hands to k_gjk_cont (NarrowArgs.contq)
  unsigned int n_pen_refiled; // pairs k_tail_pen filed under their block's shard because their class's shards were full
  unsigned int pad[20];
};
struct Counters {
  CounterShard sh[TOHC_NSHARD];
  unsigned int overflow;
  unsigned int epa_next;   // k_epa's work counter: wavefronts take the next BLOCK / G pairs when they are done with theirs
  unsigned int n_dist;     // pairs whose origin is outside (k_tail_pen) and that need the Distance fall-back (NarrowArgs.distq)
  unsigned int pad[29];
};
#define TOHC_CNT_STRIDE (sizeof(CounterShard) / sizeof(unsigned int))

struct CallParams {
  int n_cast_sets;   // number of (problem, t) units enumerated by this launch ...
  int unit_base;     // ... starting at this one (an enqueue may be cut into unit ranges)
  int T, gap;        // sets per problem; cast: second pose = set + gap (robot mode); poses mode: T = 1, gap = 0
  int cast;          // 0 discrete, 1 cast
  int filter_mask;
  int want_jac;
  float d, breaking;
  unsigned int seg_cap;      // capacity of ONE shard of the pair / EPA / hit queues
  unsigned int max_records;
  unsigned int pen_cap;      // capacity of ONE shard of the k_tail_pen -> k_tail_epa queue: seg_cap / TOHC_PEN_DIV (less only in tests: TOHC_PEN_CAP)
  int dbg_max_iter;          // diagnostics only: 1000 = the reference's gGjkMaxIter
  int ordered_pairs;         // AllVsAll: a query child meets only the children of objects with a higher number
  unsigned int epa_small_v, epa_small_f;   // k_tail_epa's first-attempt pools (<= TOHC_EPA_SMALL_V / _F; smaller only in tests)
  int gjk_split;             // k_gjk (a lane per pair) hands a run to k_gjk_cont after this many passes of the loop; 0 = never
  int t_disc;                // discrete pass: it covers the first t_disc timesteps of each problem's T staged ones (T, or T - gap when the
                             // last `gap` poses are another shard's: tohc_enqueue_linearization_window)
};

__host__ __device__ __forceinline__ Xf ldXf(const float *p) {
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
__device__ __forceinline__ Aabb ldAabb(const float *p) {
  Aabb b;
  b.lo = mk(p[0], p[1], p[2]);
  b.hi = mk(p[3], p[4], p[5]);
  return b;
}
__device__ __forceinline__ void castSets(const CallParams &cp, int cs, int &set0, int &set1) {
  int per = cp.cast ? (cp.T - cp.gap) : cp.t_disc;
  set0 = (cs / per) * cp.T + cs % per;
  set1 = set0 + (cp.cast ? cp.gap : 0);
}

// ---- sharded queues ---------------------------------------------------------------------------------------------
struct QPrefix {
  unsigned int start[TOHC_NSHARD + 1];
};
__device__ __forceinline__ void qprefix(const unsigned int *counts, unsigned int segCap, QPrefix &p) {
  unsigned int acc = 0;
#pragma unroll
  for (int i = 0; i < TOHC_NSHARD; ++i) {
    p.start[i] = acc;
    unsigned int c = counts[i * TOHC_CNT_STRIDE];
    acc += c < segCap ? c : segCap;
  }
  p.start[TOHC_NSHARD] = acc;
}
// logical index -> physical position (shard * segCap + offset)
__device__ __forceinline__ unsigned int qlocate(const QPrefix &p, unsigned int segCap, unsigned int i) {
  unsigned int sh = 0;
#pragma unroll
  for (int k = 1; k < TOHC_NSHARD; ++k) sh += (i >= p.start[k]) ? 1u : 0u;
  unsigned int st = p.start[0];
#pragma unroll
  for (int k = 1; k < TOHC_NSHARD; ++k)
    if (k == (int)sh) st = p.start[k];
  return sh * segCap + (i - st);
}

// One queue slot for every lane that wants one, with ONE atomic per wavefront: the lanes that are active here vote,
// the first of them reserves for all.  Call it with all lanes of a (possibly divergent) branch, `want` false for the
// lanes that only take part in the vote.  Blocks of the lane-per-pair kernels are single wavefronts.
__device__ __forceinline__ unsigned int waveSlot(unsigned int *ctr, bool want) {
  const unsigned long long m = __ballot(want);
  unsigned int slot = 0;
  if (want) {
    const int lane = (int)(threadIdx.x & 63);
    const int leader = (int)__builtin_ctzll(m);
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(ctr, (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
  }
  return slot;
}
// sum of `v` over the wavefront's lanes, added to *ctr with one atomic (call with every lane of the wavefront)
__device__ __forceinline__ void waveAdd(unsigned int *ctr, unsigned int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(ctr, v);
}

// the exact stage of the pre-test: the gap between the two shapes along `dir` (other child -> query origin; eA = the
// query's reach along -dir, mq its margin) is a lower bound of their distance; true = provably more than lim apart
__device__ __forceinline__ bool exactBeyond(V3 dir, float dl2, V3 oq, float eA, float mq, const Cvx<1> &cx, const Xf &tx, float lim) {
  const float eB = reachNV(cx, mulVM(dir, tx.b));
  const float delta = dot(dir, oq - tx.o) - eA - eB;
  const float limm = lim + (mq + cvxMargin<1>(cx));
  return (delta > 0.0f) && (delta * delta > (limm * limm) * dl2);
}

// the sphere stage of k_broadphase's pre-test for a child that moves: what CullSph holds for one that never does, at this pose
__device__ __forceinline__ CullSph movingSph(const DevScene &sc, int child, const Xf &T) {
  const ChildDesc *dd = sc.child_desc + child;
  const F4 sl = *reinterpret_cast<const F4 *>(dd->sph);
  const V3 w = xfApply(T, mk(sl.x, sl.y, sl.z));
  CullSph o;
  o.c[0] = w.x; o.c[1] = w.y; o.c[2] = w.z; o.r = sl.w;
  o.obj = dd->obj; o.group = dd->group;
  return o;
}

// -----------------------------------------------------------------------------------------------------------------
// k_fk: RobotAndDOF::SetDOFValues (configuration_space.cpp:12-23) + link->GetTransform() for every link, plus what
// UpdateBulletFromRave / updateAabbs do for the robot's links (bullet_collision_checker.cpp:648-653).
// -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fk(DevRobot rb, DevScene sc, DynMap dm, const double *__restrict__ dofs, int n_cfg, float d,
                                           double *__restrict__ linkQT, double *__restrict__ objQT, float *__restrict__ dynXf,
                                           float *__restrict__ dynAabb, double *__restrict__ linkZ = nullptr, CullSph *__restrict__ dynSph = nullptr) {
  extern __shared__ double s_motion[];  // [n_links][7]
  const int s = blockIdx.x;
  if (s >= n_cfg) return;
  const double *q = dofs + (size_t)s * rb.K;
  for (int i = threadIdx.x; i < rb.n_links; i += blockDim.x) {
    Dxf m;
    m.q.w = 1; m.q.x = m.q.y = m.q.z = 0;
    m.t = mkd(0, 0, 0);
    if (rb.jtype[i] != TOHC_J_FIXED) {
      double v = rb.offs[i];
      if (rb.dof[i] >= 0) {
        double qv = rb.dof_col[i] >= 0 ? q[rb.dof_col[i]] : rb.q_robot[rb.dof[i]];
        v = rb.mult[i] * qv + rb.offs[i];
      }
      D3 ax = mkd(rb.axis[3 * i], rb.axis[3 * i + 1], rb.axis[3 * i + 2]);
      if (rb.jtype[i] == TOHC_J_REVOLUTE) {
        double h = 0.5 * v, sn = sin(h);
        m.q.w = cos(h); m.q.x = ax.x * sn; m.q.y = ax.y * sn; m.q.z = ax.z * sn;
      } else {
        m.t = ax * v;
      }
    }
    dxfTo7(m, s_motion + 7 * i);
  }
  __syncthreads();
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
  for (int i = threadIdx.x; i < rb.n_links; i += blockDim.x) {
    Dxf t = base;
    for (int c = rb.chain_off[i]; c < rb.chain_off[i + 1]; ++c) {
      const int a = rb.chain[c];
      t = dxfMul(t, dxfFrom7(rb.T_origin + 7 * a));
      if (rb.jtype[a] != TOHC_J_FIXED) t = dxfMul(t, dxfFrom7(s_motion + 7 * a));
    }
    double t7[7];
    dxfTo7(t, t7);
    double *out = linkQT + ((size_t)s * rb.n_links + i) * 7;
    for (int k = 0; k < 7; ++k) out[k] = t7[k];
    if (linkZ) {   // the joint axis in the world, once per (configuration, link) instead of once per contact and joint (k_finish)
      const D3 z = qrot(t.q, mkd(rb.axis[3 * i], rb.axis[3 * i + 1], rb.axis[3 * i + 2]));
      double *oz = linkZ + ((size_t)s * rb.n_links + i) * 3;
      oz[0] = z.x; oz[1] = z.y; oz[2] = z.z;
    }
    const int slot = rb.link_dyn[i];
    if (objQT && slot >= 0) {
      double *oo = objQT + ((size_t)s * dm.n_dyn_obj + slot) * 7;
      for (int k = 0; k < 7; ++k) oo[k] = t7[k];
      // the link's geometry: fp32 pose exactly as toBt() makes it, child poses, cull boxes
      const int obj = rb.link_obj[i];
      const Xf T = xfFrom7(t7);
      for (int k = 0; k < sc.obj_cnt[obj]; ++k) {
        const int c = sc.obj_off[obj] + k;
        Xf Tc = T;
        if (sc.obj_compound[obj]) Tc = xfMul(T, ldXf(sc.child_localT + 12 * c));
        const size_t o = (size_t)s * dm.n_dyn_children + dm.child_dyn[c];
        stXf(dynXf + o * 12, Tc);
        const int sh = sc.child_shape[c];
        Aabb b = shapeAabb(mk(sc.shape_lmin[3 * sh], sc.shape_lmin[3 * sh + 1], sc.shape_lmin[3 * sh + 2]),
                           mk(sc.shape_lmax[3 * sh], sc.shape_lmax[3 * sh + 1], sc.shape_lmax[3 * sh + 2]), Tc, d);
        float *ob = dynAabb + o * 6;
        ob[0] = b.lo.x; ob[1] = b.lo.y; ob[2] = b.lo.z; ob[3] = b.hi.x; ob[4] = b.hi.y; ob[5] = b.hi.z;
        if (dynSph) dynSph[o] = movingSph(sc, c, Tc);
      }
    }
  }
}

// k_dyn_children ("poses" mode): object pose (7 doubles) -> fp32 transform exactly as toBt() does, child world
// transform (orgTrans*childTrans for compounds, btCompoundCollisionAlgorithm.cpp:122-125) and the cull box.
__global__ void k_dyn_children(DevScene sc, DynMap dm, const double *__restrict__ objQT, int n_sets, float d, float *__restrict__ dynXf,
                               float *__restrict__ dynAabb, CullSph *__restrict__ dynSph = nullptr) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)n_sets * dm.n_dyn_children;
  if (i >= total) return;
  int set = (int)(i / dm.n_dyn_children), slot = (int)(i % dm.n_dyn_children);
  int c = dm.dyn_children[slot];
  int obj = sc.child_obj[c];
  Xf T = xfFrom7(objQT + ((size_t)set * dm.n_dyn_obj + dm.obj_dyn[obj]) * 7);
  if (sc.obj_compound[obj]) T = xfMul(T, ldXf(sc.child_localT + 12 * c));
  stXf(dynXf + i * 12, T);
  int sh = sc.child_shape[c];
  Aabb b = shapeAabb(mk(sc.shape_lmin[3 * sh], sc.shape_lmin[3 * sh + 1], sc.shape_lmin[3 * sh + 2]),
                     mk(sc.shape_lmax[3 * sh], sc.shape_lmax[3 * sh + 1], sc.shape_lmax[3 * sh + 2]), T, d);
  float *o = dynAabb + i * 6;
  o[0] = b.lo.x; o[1] = b.lo.y; o[2] = b.lo.z; o[3] = b.hi.x; o[4] = b.hi.y; o[5] = b.hi.z;
  if (dynSph) dynSph[i] = movingSph(sc, c, T);
}

// -----------------------------------------------------------------------------------------------------------------
// k_ray: RayCastCollision (bullet_collision_checker.cpp:352-360) for a batch of rays.  ONE LANE per (ray, convex child):
// btCollisionWorld::rayTest visits every object (the callback's group / mask are DefaultFilter / AllFilter, the allowed-
// pair matrix is not consulted) and every child of a compound (btCollisionWorld.cpp:554-557); the
// ClosestRayResultCallback keeps the smallest fraction -- here an atomicMin on the bits of the (non-negative) fraction.
// The lanes of a wavefront share the child (its vertices arrive as one broadcast load) and take consecutive rays.
// No cull, like the oracle: Bullet's DBVT ray cull (boxes grown by 3.001 d plus history-dependent fat margins) only
// matters for the rare hit btSubsimplexConvexCast reports away from the shape (stale simplex vertices), DESIGN.md 5a.
// -----------------------------------------------------------------------------------------------------------------
struct RayArgs {
  DevScene sc;
  DynMap dm;
  const float *dynXf;      // child poses of the one pose set
  const float *rays;       // [n_rays][8]: from xyz, 0, to xyz, 0 -- already rounded to fp32 as toBt() does
  unsigned int *closest;   // [n_rays] bits of m_closestHitFraction, preset to 1.0f
  int n_rays;
};
__global__ void __launch_bounds__(256) k_ray(RayArgs ra) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)ra.n_rays * ra.sc.n_children) return;
  const int ray = (int)(i % ra.n_rays), child = (int)(i / ra.n_rays);
  const ChildDesc cd = ra.sc.child_desc[child];
  Cvx<1> cx;
  cvxLoad<1>(cx, cd.kind, mk(cd.dims[0], cd.dims[1], cd.dims[2]), ra.sc.vx, ra.sc.vy, ra.sc.vz, ra.sc.v4, cd.voff, cd.vcnt, ra.sc.v8, cd.voff8);
  const int slot = ra.dm.child_dyn[child];
  const Xf tx = slot >= 0 ? ldXf(ra.dynXf + (size_t)slot * 12) : ldXf(ra.sc.stat_xf + 12 * child);
  const F4 a = reinterpret_cast<const F4 *>(ra.rays)[2 * ray], b = reinterpret_cast<const F4 *>(ra.rays)[2 * ray + 1];
  float fr;
  V3 nrm;
  if (rayConvex(cx, tx, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), fr, nrm) && fr < 1.0f && fr >= 0.0f)
    atomicMin(ra.closest + ray, __float_as_uint(fr));
}

// rays given as doubles in device memory -> the fp32 table k_ray reads (toBt(): double -> float,
// bullet_collision_checker.cpp:72-74) + the callback's initial m_closestHitFraction = 1 (tohc_ray_cast_poses_noalloc;
// the allocating entry points do both on the host)
__global__ void __launch_bounds__(256) k_ray_prep(const double *__restrict__ from, const double *__restrict__ to, int n_rays, float *__restrict__ rays,
                                                  unsigned int *__restrict__ closest) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rays) return;
  F4 a, b;
  a.x = (float)from[3 * i]; a.y = (float)from[3 * i + 1]; a.z = (float)from[3 * i + 2]; a.w = 0.0f;
  b.x = (float)to[3 * i]; b.y = (float)to[3 * i + 1]; b.z = (float)to[3 * i + 2]; b.w = 0.0f;
  reinterpret_cast<F4 *>(rays)[2 * i] = a;
  reinterpret_cast<F4 *>(rays)[2 * i + 1] = b;
  closest[i] = 0x3f800000u;
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
  const double *linkZ;    // [n_sets][n_links][3]: every link's joint axis in the world (k_fk), null in poses mode
  const double *dofs;     // [n_sets][K]
  PairRec *pairs;
  PairRec *cands;         // k_broadphase -> k_exact: what the sphere stage of the pre-test passes (pad = the other child's slot + 1 if it moves)
  EpaRec *epaq;
  PenRec *penq;
  GjkCont *contq;
  unsigned int *distq;    // positions in epaq of the pairs for the Distance fall-back (k_tail_pen -> the extra blocks of k_tail_epa)
  HitRec *hits;
  Counters *cnt;
  unsigned int *n_records;   // records the slab holds once every enqueue so far has finished -- advanced by k_range
  const unsigned int *rec_base;   // where this enqueue's records start (k_range wrote it before k_finish runs)
  tohc_record *records;
  double *grads;             // [max_records][2][K]
  unsigned long long *dbg;   // diagnostics (TOHC_EPA_PROF builds), else null
};

struct PairIds {
  int set0, set1, Q, qc, X, xc;
  bool xIsA;  // X's child is GJK's A (X is a compound), else the query child is A (dispatch roles)
};
__device__ __forceinline__ void pairIds(const NarrowArgs &a, const PairRec &p, PairIds &c) {
  castSets(a.cp, p.set, c.set0, c.set1);
  c.Q = a.sc.child_obj[p.q]; c.X = a.sc.child_obj[p.x];
  c.qc = a.sc.child_idx[p.q]; c.xc = a.sc.child_idx[p.x];
  c.xIsA = a.sc.obj_compound[c.X] != 0;
}
template <int G>
__device__ __forceinline__ void loadShape(const DevScene &sc, int child, Cvx<G> &c) {
  const int sh = sc.child_shape[child];
  cvxLoad<G>(c, sc.shape_kind[sh], mk(sc.shape_dims[3 * sh], sc.shape_dims[3 * sh + 1], sc.shape_dims[3 * sh + 2]), sc.vx, sc.vy, sc.vz, sc.v4,
             sc.shape_voff[sh], sc.shape_vcnt[sh]);
}
__device__ __forceinline__ Xf childXf(const NarrowArgs &a, const float *dyn, int set, int child) {
  const int s = a.dm.child_dyn[child];
  return s >= 0 ? ldXf(dyn + ((size_t)set * a.dm.n_dyn_children + s) * 12) : ldXf(a.sc.stat_xf + 12 * child);
}
// one side of a pair: shape into registers + its world pose(s)
template <int G>
// (slot: the child's slot in the pose tables when the caller knows it, -1 if it never moves, -2 = look it up)
__device__ __forceinline__ void loadSide(const NarrowArgs &a, int child, const ChildDesc &cd, int set0, int set1, bool cast, Cvx<G> &c, Xf &t0, int slot = -2) {
  cvxLoad<G>(c, cd.kind, mk(cd.dims[0], cd.dims[1], cd.dims[2]), a.sc.vx, a.sc.vy, a.sc.vz, a.sc.v4, cd.voff, cd.vcnt, a.sc.v8, cd.voff8);
  if (slot == -2) slot = a.dm.child_dyn[child];
  t0 = slot >= 0 ? ldXf(a.dynXf0 + ((size_t)set0 * a.dm.n_dyn_children + slot) * 12) : ldXf(a.sc.stat_xf + 12 * child);
  if (cast) {
    Xf t1 = slot >= 0 ? ldXf(a.dynXf1 + ((size_t)set1 * a.dm.n_dyn_children + slot) * 12) : ldXf(a.sc.stat_xf + 12 * child);
    c.cast = 1;
    c.t01 = xfInverseTimes(t0, t1);  // CastHullShape(convex, tf0.inverseTimes(tf1)), bullet_collision_checker.cpp:977
  }
}

// The same for a child that is the same in every lane (k_broadphase's query): descriptor and poses through scalar loads
__device__ __forceinline__ Xf ldXfU(const float *p) {
  const TOHC_CONST_AS float *u = uniformPtr(p);
  Xf t;
  t.b.r[0] = mk(u[0], u[1], u[2]);
  t.b.r[1] = mk(u[3], u[4], u[5]);
  t.b.r[2] = mk(u[6], u[7], u[8]);
  t.o = mk(u[9], u[10], u[11]);
  return t;
}
__device__ __forceinline__ Aabb ldAabbU(const float *p) {
  const TOHC_CONST_AS float *u = uniformPtr(p);
  Aabb b;
  b.lo = mk(u[0], u[1], u[2]);
  b.hi = mk(u[3], u[4], u[5]);
  return b;
}
__device__ __forceinline__ void loadSideU(const NarrowArgs &a, int child, int set0, int set1, bool cast, Cvx<1> &c, Xf &t0, int slot) {
  const TOHC_CONST_AS ChildDesc *cd = uniformPtr(a.sc.child_desc + child);
  c.kind = cd->kind;
  c.cast = 0;
  c.n = cd->vcnt;
  c.dims = mk(cd->dims[0], cd->dims[1], cd->dims[2]);
  c.vb = a.sc.v8 + (size_t)cd->voff8 * TOHC_VBLOCK;
  t0 = slot >= 0 ? ldXfU(a.dynXf0 + ((size_t)set0 * a.dm.n_dyn_children + slot) * 12) : ldXfU(a.sc.stat_xf + 12 * child);
  if (cast) {
    Xf t1 = slot >= 0 ? ldXfU(a.dynXf1 + ((size_t)set1 * a.dm.n_dyn_children + slot) * 12) : ldXfU(a.sc.stat_xf + 12 * child);
    c.cast = 1;
    c.t01 = xfInverseTimes(t0, t1);
  }
}

// k_cast_boxes: the second box of a swept child, as the reference makes it.  CastHullShape::getAabb
// (bullet_collision_checker.cpp:850-856) joins the shape's box at t with its box at t * m_t01, and m_t01 = tf0^-1 * tf1 is
// rounded: t * m_t01 is tf1 to an ulp or two, not tf1.  The box of the pose table's tf1 (k_fk / k_dyn_children) differs
// from it by as little -- and flipped one box test in 22.6 million on config 4 (a pair evaluation the oracle counted and
// the engine did not; no contact can hang on it, both boxes are grown by d).  One thread per (cast unit, moving child):
// out[set1][slot] = box at tf0 * (tf0^-1 * tf1), read by the broadphase kernels in place of the table's box at set1.
__global__ void __launch_bounds__(256) k_cast_boxes(DevScene sc, DynMap dm, CallParams cp, const float *__restrict__ dynXf0, const float *__restrict__ dynXf1,
                                                    float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t ndc = (size_t)dm.n_dyn_children;
  if (i >= (size_t)cp.n_cast_sets * ndc) return;
  const int cs = cp.unit_base + (int)(i / ndc), slot = (int)(i % ndc);
  int set0, set1;
  castSets(cp, cs, set0, set1);
  const Xf t0 = ldXf(dynXf0 + ((size_t)set0 * ndc + slot) * 12), t1 = ldXf(dynXf1 + ((size_t)set1 * ndc + slot) * 12);
  const Xf tw1 = xfMul(t0, xfInverseTimes(t0, t1));
  const int sh = sc.child_shape[dm.dyn_children[slot]];
  const Aabb b = shapeAabb(mk(sc.shape_lmin[3 * sh], sc.shape_lmin[3 * sh + 1], sc.shape_lmin[3 * sh + 2]),
                           mk(sc.shape_lmax[3 * sh], sc.shape_lmax[3 * sh + 1], sc.shape_lmax[3 * sh + 2]), tw1, cp.d);
  float *ob = out + ((size_t)set1 * ndc + slot) * 6;
  ob[0] = b.lo.x; ob[1] = b.lo.y; ob[2] = b.lo.z; ob[3] = b.hi.x; ob[4] = b.hi.y; ob[5] = b.hi.z;
}

// -----------------------------------------------------------------------------------------------------------------
// k_broadphase: ONE WAVEFRONT per (unit, query child).  Filters follow CollisionCollector::needsCollision
// (bullet_collision_checker.cpp:409-413) and btSingleContactCallback::process (btCollisionWorld.cpp:1129-1137); the
// cull is the engine's own conservative box test (same arithmetic as oracle/trajopt_oracle.c shapeAabb): both boxes
// grown by d, closed-interval overlap -- it can never drop a pair whose distance is <= d.
//   * children that never move sit in a Morton-ordered SoA table cut into clusters of 64 with a box per cluster
//     (built on the host, refreshStatic): lane i tests cluster box i, then the wave scans only the clusters that
//     overlap, 64 boxes per coalesced load -- the cull result is the same set as a full scan would give;
//   * moving children (robot links; everything in "poses" mode) are scanned directly;
//   * every box-test survivor is one PAIR EVALUATION (n_cand, what tohc_counters.pair_evals reports, equal to the
//     oracle's count).  The evaluation starts right here, one lane per pair: one support query along the line between
//     the two origins bounds the distance from below (sepAxisBeyond); pairs provably farther apart than d are
//     finished, the rest go to k_gjk through the pair list (one atomic per drained queue, sharded).
// -----------------------------------------------------------------------------------------------------------------
#define TOHC_BP_THREADS 64
#define TOHC_BP_QCAP 256
#ifndef TOHC_BP_EXACT
#define TOHC_BP_EXACT 1   // 0: no exact stage here; what the sphere stage passes goes to k_gjk, whose first iterations end far pairs too
#endif
#define TOHC_BP_DYN 0x40000000   // queue entries: child number, + this bit when the child moves (came from the scan of the moving children)
struct CullArgs {
  NarrowArgs a;
  const int *qchildren;
  int nQ;
  const float *dynAabb0, *dynAabb1;
  const CullSph *dynSph0;   // [n_sets][moving children]: bounding spheres at the (first) pose of each set
  int pretest;
  int dyn_elsewhere;   // k_broadphase: the moving children are k_broadphase_unit's; k_broadphase_unit: 1 = it also takes the static ones
  int defer_exact;     // k_broadphase: the exact stage of the pre-test runs in k_exact, on full wavefronts (NarrowArgs.cands)
  int dbg;   // timing experiments only (TOHC_BP_DBG): 1 = count the box-test survivors and drop them, 2 = stage 1 only, drop the rest
};
// DEFER: the exact stage of the pre-test is k_exact's (ca.defer_exact), and this kernel holds no code for it
template <bool CAST, bool DEFER>
#ifndef TOHC_BP_WAVES
#define TOHC_BP_WAVES 5   // <= 96 VGPRs: five resident waves per SIMD instead of four hide more of the load chain (6: spills, slower)
#endif
__global__ void __launch_bounds__(TOHC_BP_THREADS, TOHC_BP_WAVES) k_broadphase(CullArgs ca) {
  const NarrowArgs &a = ca.a;
  const DevScene &sc = a.sc;
  const DynMap &dm = a.dm;
  const CallParams &cp = a.cp;
  __shared__ int s_q[TOHC_BP_QCAP];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x % ca.nQ;
  const int cs = cp.unit_base + blockIdx.x / ca.nQ;
  // everything about the query child is the same in every lane: scalar loads (uniformPtr)
  const TOHC_CONST_AS int *qinfo = uniformPtr(ca.qchildren + 4 * qi);   // (child, object, moving slot or -1, 0)
  const int q = qinfo[0];
  const int Q = qinfo[1];
  const int shard = blockIdx.x % TOHC_NSHARD;
  int set0, set1;
  castSets(cp, cs, set0, set1);
  const int qs = qinfo[2];
  Aabb bq = qs >= 0 ? ldAabbU(ca.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + qs) * 6) : ldAabbU(sc.stat_aabb + 6 * q);
  if (CAST && qs >= 0) bq = aabbUnion(bq, ldAabbU(ca.dynAabb1 + ((size_t)set1 * dm.n_dyn_children + qs) * 6));
  const unsigned char *allowedRow = sc.allowed + (size_t)Q * sc.n_objects;
  const float lim = cp.d * 1.01f + 1.0e-4f;  // guard band: the bound and Bullet's result both carry ~1e-6 of rounding
  unsigned int nq = 0, ncand = 0;            // wave-uniform
  // the query side of the pre-test (descriptor -> pose(s)): issued here so that these dependent loads travel together
  // with the scan's instead of starting a chain of their own at the first drain
  Cvx<1> cq;
  Xf tq;
  if (ca.pretest) loadSideU(a, q, set0, set1, CAST, cq, tq, qs);
  // first block of the moving children's boxes, likewise ahead of the scan of the static table
  Aabb bd0;
  bd0.lo = mk(1.f, 1.f, 1.f); bd0.hi = mk(-1.f, -1.f, -1.f);
  if (lane < dm.n_dyn_children && ca.dbg != 4 && !ca.dyn_elsewhere) bd0 = ldAabb(ca.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + lane) * 6);

  // The queued box-test survivors, one lane per pair: object-level filters (here and not in the scan loop: they are
  // three dependent loads, which per overlapping cluster made the scan a chain of ~100 L2 round trips per wavefront),
  // pre-test, and what is left goes to the pair list.
  auto drain = [&]() __attribute__((always_inline)) {
    __syncthreads();   // one wavefront per block: orders the queue writes before the reads below
    for (unsigned int k0 = 0; k0 < nq; k0 += TOHC_BP_THREADS) {
      const unsigned int k = k0 + lane;
      bool keep = k < nq;
      int x = 0, xSlot = 0;   // the other child's number; for one that moves also its slot in the pose tables
      bool xDyn = false;
      V3 c = mk(0.f, 0.f, 0.f);   // the other child's bounding sphere, world coordinates
      float rad = 0.f;
      if (keep) {
        const int ent = s_q[k];
        x = ent & ~TOHC_BP_DYN;
        xDyn = (ent & TOHC_BP_DYN) != 0;
        xSlot = x;
        // one 32-byte record either way: the child's bounding sphere in the world + its object / group words (a child that
        // never moves: made on the host with the static poses; one that moves: by k_fk / k_dyn_children for this set --
        // queue entries of moving children carry their slot, not their number)
        const CullSph sx = xDyn ? ca.dynSph0[(size_t)set0 * dm.n_dyn_children + x] : sc.stat_sph[x];
        if (xDyn) x = dm.dyn_children[x];
        const int X = sx.obj, gx = sx.group;
        c = mk(sx.c[0], sx.c[1], sx.c[2]);
        rad = sx.r;
        // discrete: contactTest skips the query object itself (btCollisionWorld.cpp:1132-1133).  Cast: the swept shape is a
        // temporary object that is not in the world, so nothing is skipped -- a query link whose own object passes the
        // KinBodyFilter mask (a grabbed body's link) meets its resting self, as in the reference (bcc:983, 886-892)
        keep = (CAST || X != Q) && (CAST ? (gx & TOHC_GROUP_KINBODY) != 0 : (gx & (short)cp.filter_mask) != 0) && allowedRow[X] != 0 &&
               (!cp.ordered_pairs || X > Q);
      }
      ncand += (unsigned int)__popcll(__ballot(keep));
      if (ca.dbg == 1) keep = false;
      if (keep && ca.pretest) {
        // one direction for both stages: from the other child's bounding-sphere centre to the query's origin.
        // Stage 1 needs nothing of the other child but that sphere: the query's support point against the SPHERE bounds
        // the distance from below; ~94 % of what the exact test rejects ends here.
        // Stage 2 (the exact support of the other child, a scan of its vertices) only runs for the rest.
        const V3 dir = tq.o - c;
        const float dl2 = len2(dir);
        if (dl2 > 1.0e-12f) {
          const float dl = sqrtf(dl2);
          const float eA = reachNV<true>(cq, mulVM(-dir, tq.b));   // how far the query reaches towards the other child
          const float mq = cvxMargin<1>(cq);   // a plain sphere is its centre + this much (0 for everything else)
          const float deltaS = dl2 - eA - (rad + mq) * dl;
          if ((deltaS > 0.0f && deltaS > lim * dl) || ca.dbg == 2) {
            keep = false;
          } else if (TOHC_BP_EXACT && !DEFER) {
            const ChildDesc dx = sc.child_desc[x];
            // (a child that never moves: its pose straight from the static table, no look-up of its slot first)
            const Xf tx = xDyn ? ldXf(a.dynXf0 + ((size_t)set0 * dm.n_dyn_children + xSlot) * 12) : ldXf(sc.stat_xf + 12 * x);
            Cvx<1> cx;
            cvxLoad<1>(cx, dx.kind, mk(dx.dims[0], dx.dims[1], dx.dims[2]), sc.vx, sc.vy, sc.vz, sc.v4, dx.voff, dx.vcnt, sc.v8, dx.voff8);
            keep = !exactBeyond(dir, dl2, tq.o, eA, mq, cx, tx, lim);
          }
        }
      }
      // survivors: to k_gjk's pair list -- or, with the exact stage deferred, to k_exact's list: only one lane in ten gets
      // this far, and run here the exact stage is ~250 instructions of every wavefront for the four or five lanes that
      // need it; k_exact runs it on full wavefronts
      const unsigned long long m = __ballot(keep);
      const unsigned int cnt = (unsigned int)__popcll(m);
      if (cnt > 0) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(DEFER ? &a.cnt->sh[shard].n_exact : &a.cnt->sh[shard].n_pairs, cnt);
        base = __shfl(base, 0);
        if (keep) {
          const unsigned int pos = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < cp.seg_cap) {
            PairRec p;
            p.set = cs; p.q = q; p.x = x; p.pad = xDyn ? xSlot + 1 : 0;
            (DEFER ? a.cands : a.pairs)[(size_t)shard * cp.seg_cap + pos] = p;
          } else {
            a.cnt->overflow = 1;
          }
        }
      }
    }
    __syncthreads();
    nq = 0;
  };
  auto push = [&](bool hit, int x) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(hit);
    if (hit) s_q[nq + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = x;
    nq += (unsigned int)__popcll(m);
    if (nq > TOHC_BP_QCAP - TOHC_BP_THREADS) drain();
  };

  // children that never move: cluster boxes first, then the members of the clusters that overlap
  for (int cb = 0; cb < (ca.dbg == 3 ? 0 : dm.n_cl); cb += TOHC_BP_THREADS) {
    const int c = cb + lane;
    bool over = false;
    if (c < dm.n_cl) {
      const float *t = dm.cl_soa + c;
      const int P = dm.n_cl_pad;
      Aabb bc;
      bc.lo = mk(t[0], t[P], t[2 * P]);
      bc.hi = mk(t[3 * P], t[4 * P], t[5 * P]);
      over = aabbOverlap(bq, bc);
    }
    unsigned long long cm = __ballot(over);
    // four clusters per round, all their loads (boxes and child numbers) issued before the first test: the loop is a
    // chain of L2 round trips otherwise (one per overlapping cluster, two with the child number loaded after the test)
    while (cm) {
      int cc[4];
      float bb[4][6];
      int ch[4];
      const int P = dm.n_stat_pad;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        cc[u] = -1;
        if (cm) { cc[u] = cb + (int)__builtin_ctzll(cm); cm &= cm - 1; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (cc[u] >= 0) {
          const int i = cc[u] * 64 + lane;       // pad entries hold inverted boxes: they never overlap
          const float *t = dm.stat_soa + i;
#pragma unroll
          for (int k = 0; k < 6; ++k) bb[u][k] = t[k * P];
          ch[u] = dm.stat_children[i];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (cc[u] >= 0) {
          Aabb bx;
          bx.lo = mk(bb[u][0], bb[u][1], bb[u][2]);
          bx.hi = mk(bb[u][3], bb[u][4], bb[u][5]);
          const bool hit = aabbOverlap(bq, bx);
          push(hit, hit ? ch[u] : 0);
        }
      }
    }
  }
  // moving children (the first block was loaded at the top)
  if (ca.dbg != 4 && !ca.dyn_elsewhere && dm.n_dyn_children > 0) push(lane < dm.n_dyn_children && aabbOverlap(bq, bd0), lane | TOHC_BP_DYN);
  for (int base = TOHC_BP_THREADS; base < (ca.dbg == 4 || ca.dyn_elsewhere ? 0 : dm.n_dyn_children); base += TOHC_BP_THREADS) {
    const int i = base + lane;
    bool hit = false;
    if (i < dm.n_dyn_children) hit = aabbOverlap(bq, ldAabb(ca.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + i) * 6));
    push(hit, i | TOHC_BP_DYN);
  }
  if (nq > 0) drain();
  if (lane == 0 && ncand > 0) atomicAdd(&a.cnt->sh[shard].n_cand, ncand);
}

// -----------------------------------------------------------------------------------------------------------------
// k_broadphase_both: the broadphase of a linearization's discrete AND cast pass in one launch (tohc_enqueue_linearization).
// The cast pass's query box (the boxes at t and t + gap joined) holds the discrete pass's (the box at t); the sphere
// stage of both pre-tests looks from the other child's sphere to the query's origin at t, and the swept shape's reach
// is the larger of the reaches at the two poses, the first of which is the discrete pass's.  So one wavefront per
// (discrete unit, query child) scans once with the larger box, notes per survivor whether the smaller box overlaps
// too, drains once and feeds both passes' candidate lists (d = the discrete pass's arguments, c = the cast pass's; each
// with its own counters and list; ONLY with the exact stage deferred, unit ranges whole, same query children).  Per pass
// the survivors, counts and lists are what k_broadphase<false> / <true> produce: same tests, same arithmetic.
// -----------------------------------------------------------------------------------------------------------------
struct BothArgs { CullArgs d, c; };
#define TOHC_BP_DFLAG 0x20000000   // queue entries: the discrete pass's (smaller) box overlaps too
__global__ void __launch_bounds__(TOHC_BP_THREADS, TOHC_BP_WAVES) k_broadphase_both(BothArgs ba) {
  const CullArgs &cd = ba.d, &cc = ba.c;
  const NarrowArgs &a = cd.a;   // (scene, maps, pose tables: the same in both)
  const DevScene &sc = a.sc;
  const DynMap &dm = a.dm;
  const CallParams &cpd = cd.a.cp, &cpc = cc.a.cp;
  __shared__ int s_q[TOHC_BP_QCAP];
  const int lane = threadIdx.x;
  const int qi = blockIdx.x % cd.nQ;
  const int u = blockIdx.x / cd.nQ;          // the discrete pass's unit: timestep t < t_disc of a problem (t_disc = T, or T - gap: then every unit has a cast unit)
  const int T = cpd.T, gap = cpc.gap;
  const int prob = u / cpd.t_disc, t = u % cpd.t_disc;
  const bool hasCast = t < T - gap;          // wave-uniform
  const int csC = prob * (T - gap) + t;      // the cast pass's unit
  const int set0 = prob * T + t, set1 = set0 + gap;
  const TOHC_CONST_AS int *qinfo = uniformPtr(cd.qchildren + 4 * qi);   // (child, object, moving slot or -1, 0)
  const int q = qinfo[0], Q = qinfo[1], qs = qinfo[2];
  const int shard = blockIdx.x % TOHC_NSHARD;
  const Aabb bqD = qs >= 0 ? ldAabbU(cd.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + qs) * 6) : ldAabbU(sc.stat_aabb + 6 * q);
  Aabb bq = bqD;
  if (hasCast && qs >= 0) bq = aabbUnion(bqD, ldAabbU(cc.dynAabb1 + ((size_t)set1 * dm.n_dyn_children + qs) * 6));
  const unsigned char *allowedRow = sc.allowed + (size_t)Q * sc.n_objects;
  const float lim = cpd.d * 1.01f + 1.0e-4f;
  unsigned int nq = 0, ncandD = 0, ncandC = 0;   // wave-uniform
  Cvx<1> cq;
  Xf tq;
  loadSideU(a, q, set0, set1, hasCast, cq, tq, qs);   // (cq.cast / cq.t01 only with a cast unit)
  Aabb bd0;
  bd0.lo = mk(1.f, 1.f, 1.f); bd0.hi = mk(-1.f, -1.f, -1.f);
  if (lane < dm.n_dyn_children) bd0 = ldAabb(cd.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + lane) * 6);

  auto drain = [&]() __attribute__((always_inline)) {
    __syncthreads();
    for (unsigned int k0 = 0; k0 < nq; k0 += TOHC_BP_THREADS) {
      const unsigned int k = k0 + lane;
      bool keepD = false, keepC = false;
      int x = 0, xSlot = 0;
      bool xDyn = false;
      V3 c = mk(0.f, 0.f, 0.f);
      float rad = 0.f;
      if (k < nq) {
        const int ent = s_q[k];
        x = ent & ~(TOHC_BP_DYN | TOHC_BP_DFLAG);
        xDyn = (ent & TOHC_BP_DYN) != 0;
        xSlot = x;
        const CullSph sx = xDyn ? cd.dynSph0[(size_t)set0 * dm.n_dyn_children + x] : sc.stat_sph[x];
        if (xDyn) x = dm.dyn_children[x];
        const int X = sx.obj, gx = sx.group;
        c = mk(sx.c[0], sx.c[1], sx.c[2]);
        rad = sx.r;
        const bool allowed = allowedRow[X] != 0;
        // (k_broadphase's filters: the discrete pass skips the query object itself, the swept shape meets everything)
        keepD = (ent & TOHC_BP_DFLAG) != 0 && X != Q && (gx & (short)cpd.filter_mask) != 0 && allowed;
        keepC = hasCast && (gx & TOHC_GROUP_KINBODY) != 0 && allowed;
      }
      ncandD += (unsigned int)__popcll(__ballot(keepD));
      ncandC += (unsigned int)__popcll(__ballot(keepC));
      if (keepD || keepC) {
        const V3 dir = tq.o - c;
        const float dl2 = len2(dir);
        if (dl2 > 1.0e-12f) {
          const float dl = sqrtf(dl2);
          const V3 dL = mulVM(-dir, tq.b);
          // the plain shape's reach at the first pose: the discrete pass's, and (a sphere's radius aside) the first of the swept shape's two
          const float eD = plainReach<true>(cq, dL, false);
          if (keepD) {
            const float deltaS = dl2 - eD - (rad + shapeMargin<1>(cq)) * dl;
            if (deltaS > 0.0f && deltaS > lim * dl) keepD = false;
          }
          if (keepC) {
            const float e0 = cq.kind == KIND_SPHERE ? plainReach<true>(cq, dL, true) : eD;
            const float e1 = plainReach<true>(cq, mulVM(dL, cq.t01.b), true) + dot(dL, cq.t01.o);
            const float deltaS = dl2 - fmaxf(e0, e1) - rad * dl;
            if (deltaS > 0.0f && deltaS > lim * dl) keepC = false;
          }
        }
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const bool keep = pass ? keepC : keepD;
        const NarrowArgs &ap = pass ? cc.a : cd.a;
        const unsigned long long m = __ballot(keep);
        const unsigned int cnt = (unsigned int)__popcll(m);
        if (cnt > 0) {
          unsigned int base = 0;
          if (lane == 0) base = atomicAdd(&ap.cnt->sh[shard].n_exact, cnt);
          base = __shfl(base, 0);
          if (keep) {
            const unsigned int pos = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < ap.cp.seg_cap) {
              PairRec p;
              p.set = pass ? csC : u; p.q = q; p.x = x; p.pad = xDyn ? xSlot + 1 : 0;
              ap.cands[(size_t)shard * ap.cp.seg_cap + pos] = p;
            } else {
              ap.cnt->overflow = 1;
            }
          }
        }
      }
    }
    __syncthreads();
    nq = 0;
  };
  auto push = [&](bool hit, int ent) __attribute__((always_inline)) {
    const unsigned long long m = __ballot(hit);
    if (hit) s_q[nq + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ent;
    nq += (unsigned int)__popcll(m);
    if (nq > TOHC_BP_QCAP - TOHC_BP_THREADS) drain();
  };

  for (int cb = 0; cb < dm.n_cl; cb += TOHC_BP_THREADS) {
    const int c = cb + lane;
    bool over = false;
    if (c < dm.n_cl) {
      const float *tt = dm.cl_soa + c;
      const int P = dm.n_cl_pad;
      Aabb bc;
      bc.lo = mk(tt[0], tt[P], tt[2 * P]);
      bc.hi = mk(tt[3 * P], tt[4 * P], tt[5 * P]);
      over = aabbOverlap(bq, bc);
    }
    unsigned long long cm = __ballot(over);
    while (cm) {
      int ccn[4];
      float bb[4][6];
      int ch[4];
      const int P = dm.n_stat_pad;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        ccn[v] = -1;
        if (cm) { ccn[v] = cb + (int)__builtin_ctzll(cm); cm &= cm - 1; }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (ccn[v] >= 0) {
          const int i = ccn[v] * 64 + lane;
          const float *tt = dm.stat_soa + i;
#pragma unroll
          for (int k = 0; k < 6; ++k) bb[v][k] = tt[k * P];
          ch[v] = dm.stat_children[i];
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (ccn[v] >= 0) {
          Aabb bx;
          bx.lo = mk(bb[v][0], bb[v][1], bb[v][2]);
          bx.hi = mk(bb[v][3], bb[v][4], bb[v][5]);
          const bool hit = aabbOverlap(bq, bx);
          push(hit, hit ? (ch[v] | (aabbOverlap(bqD, bx) ? TOHC_BP_DFLAG : 0)) : 0);
        }
      }
    }
  }
  if (dm.n_dyn_children > 0)
    push(lane < dm.n_dyn_children && aabbOverlap(bq, bd0), lane | TOHC_BP_DYN | (aabbOverlap(bqD, bd0) ? TOHC_BP_DFLAG : 0));
  for (int base = TOHC_BP_THREADS; base < dm.n_dyn_children; base += TOHC_BP_THREADS) {
    const int i = base + lane;
    bool hit = false, hitD = false;
    if (i < dm.n_dyn_children) {
      const Aabb bx = ldAabb(cd.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + i) * 6);
      hit = aabbOverlap(bq, bx);
      hitD = aabbOverlap(bqD, bx);
    }
    push(hit, i | TOHC_BP_DYN | (hitD ? TOHC_BP_DFLAG : 0));
  }
  if (nq > 0) drain();
  if (lane == 0 && ncandD > 0) atomicAdd(&cd.a.cnt->sh[shard].n_cand, ncandD);
  if (lane == 0 && ncandC > 0) atomicAdd(&cc.a.cnt->sh[shard].n_cand, ncandC);
}

// -----------------------------------------------------------------------------------------------------------------
// k_exact: the exact stage of k_broadphase's pre-test, ONE LANE per pair the sphere stage passed (NarrowArgs.cands).
// Same direction (query origin - the other child's bounding-sphere centre, read from the same record), same support
// points, same test as the stage inside the broadphase kernels: the pair list k_gjk gets holds the same pairs.
// -----------------------------------------------------------------------------------------------------------------
#ifndef TOHC_EXACT_GRID
#define TOHC_EXACT_GRID 4096
#endif
template <bool CAST>
__global__ void __launch_bounds__(64) k_exact(CullArgs ca) {
  const NarrowArgs &a = ca.a;
  const DevScene &sc = a.sc;
  const DynMap &dm = a.dm;
  const CallParams &cp = a.cp;
  const int lane = threadIdx.x;
  const int shard = blockIdx.x % TOHC_NSHARD;
  QPrefix pre;
  qprefix(&a.cnt->sh[0].n_exact, cp.seg_cap, pre);
  const unsigned int n = pre.start[TOHC_NSHARD];
  const float lim = cp.d * 1.01f + 1.0e-4f;
  for (unsigned int i0 = blockIdx.x * 64u; i0 < n; i0 += gridDim.x * 64u) {   // (wave-uniform)
    const unsigned int i = i0 + lane;
    bool keep = i < n;
    PairRec p;
    p.set = p.q = p.x = p.pad = 0;
    if (keep) {
      p = a.cands[qlocate(pre, cp.seg_cap, i)];
      int set0, set1;
      castSets(cp, p.set, set0, set1);
      const int xSlot = p.pad - 1;
      const CullSph sx = xSlot >= 0 ? ca.dynSph0[(size_t)set0 * dm.n_dyn_children + xSlot] : sc.stat_sph[p.x];
      const ChildDesc dx = sc.child_desc[p.x];
      const Xf tx = xSlot >= 0 ? ldXf(a.dynXf0 + ((size_t)set0 * dm.n_dyn_children + xSlot) * 12) : ldXf(sc.stat_xf + 12 * p.x);
      Cvx<1> cq;
      Xf tq;
      loadSide<1>(a, p.q, sc.child_desc[p.q], set0, set1, CAST, cq, tq);
      const V3 dir = tq.o - mk(sx.c[0], sx.c[1], sx.c[2]);
      const float dl2 = len2(dir);
      if (dl2 > 1.0e-12f) {
        Cvx<1> cx;
        cvxLoad<1>(cx, dx.kind, mk(dx.dims[0], dx.dims[1], dx.dims[2]), sc.vx, sc.vy, sc.vz, sc.v4, dx.voff, dx.vcnt, sc.v8, dx.voff8);
        keep = !exactBeyond(dir, dl2, tq.o, reachNV(cq, mulVM(-dir, tq.b)), cvxMargin<1>(cq), cx, tx, lim);
      }
    }
    const unsigned int pos = waveSlot(&a.cnt->sh[shard].n_pairs, keep);
    if (keep) {
      if (pos < cp.seg_cap) {
        p.pad = 0;
        a.pairs[(size_t)shard * cp.seg_cap + pos] = p;
      } else {
        a.cnt->overflow = 1;
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_broadphase_unit: ONE WAVEFRONT per unit for everything a short list of boxes has to offer -- the moving children
// (a robot's own links: 45 boxes on config 4, where a wavefront per (unit, query child) spent a quarter of its time on
// one under-filled round of them and its set-up), and in a small world (a table, a kitchen: the multi-start
// configuration) the static children as well, in which case k_broadphase is not launched at all.  The wavefront walks
// the unit's query children; every round tests the query's box against 64 of the list's; the survivors of ALL the
// unit's queries share one LDS queue, so the drains run on full wavefronts.  A queue entry names its query, and the
// pre-test loads the query side per lane.  Filters, pre-test and counting are k_broadphase's.
// -----------------------------------------------------------------------------------------------------------------
#define TOHC_BPU_QCAP 512
#define TOHC_BPU_DYN (1u << 22)      // queue entries: query number << 23 | this bit for a moving child (then its slot) | child number
#define TOHC_BPU_MAX_Q 512
#define TOHC_BPU_MAX_CHILD (1u << 22)
template <bool CAST>
__global__ void __launch_bounds__(TOHC_BP_THREADS, TOHC_BP_WAVES) k_broadphase_unit(CullArgs ca) {
  const NarrowArgs &a = ca.a;
  const DevScene &sc = a.sc;
  const DynMap &dm = a.dm;
  const CallParams &cp = a.cp;
  __shared__ unsigned int s_q[TOHC_BPU_QCAP];
  const int lane = threadIdx.x;
  const int cs = cp.unit_base + blockIdx.x;
  const int shard = blockIdx.x % TOHC_NSHARD;
  int set0, set1;
  castSets(cp, cs, set0, set1);
  const float lim = cp.d * 1.01f + 1.0e-4f;
  unsigned int nq = 0, ncand = 0;   // wave-uniform
  const int nStat = (ca.dyn_elsewhere && dm.n_stat > 0) ? dm.n_stat_pad : 0;   // ("poses" mode: no static table at all)
  const int nO = nStat + dm.n_dyn_children;

  auto drain = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the queue is this wavefront's own: its LDS operations are in order
    for (unsigned int k0 = 0; k0 < nq; k0 += TOHC_BP_THREADS) {
      const unsigned int k = k0 + lane;
      bool keep = k < nq;
      int x = 0, xSlot = 0, q = 0, qs = -1;
      bool xDyn = false;
      V3 c = mk(0.f, 0.f, 0.f);
      float rad = 0.f;
      if (keep) {
        const unsigned int ent = s_q[k];
        const int4 qinfo = reinterpret_cast<const int4 *>(ca.qchildren)[ent >> 23];   // (child, object, moving slot or -1, 0)
        q = qinfo.x; qs = qinfo.z;
        const int Q = qinfo.y;
        x = (int)(ent & (TOHC_BPU_DYN - 1u));
        xDyn = (ent & TOHC_BPU_DYN) != 0;
        xSlot = x;
        const CullSph sx = xDyn ? ca.dynSph0[(size_t)set0 * dm.n_dyn_children + x] : sc.stat_sph[x];
        if (xDyn) x = dm.dyn_children[x];
        const int X = sx.obj, gx = sx.group;
        c = mk(sx.c[0], sx.c[1], sx.c[2]);
        rad = sx.r;
        // (k_broadphase's filters: contactTest skips the query object itself, the swept shape of a cast meets everything)
        keep = (CAST || X != Q) && (CAST ? (gx & TOHC_GROUP_KINBODY) != 0 : (gx & (short)cp.filter_mask) != 0) &&
               sc.allowed[(size_t)Q * sc.n_objects + X] != 0 && (!cp.ordered_pairs || X > Q);
      }
      ncand += (unsigned int)__popcll(__ballot(keep));
      if (ca.dbg == 1) keep = false;
      if (keep && ca.pretest) {
        Cvx<1> cq;
        Xf tq;
        loadSide<1>(a, q, sc.child_desc[q], set0, set1, CAST, cq, tq, qs);
        const V3 dir = tq.o - c;
        const float dl2 = len2(dir);
        if (dl2 > 1.0e-12f) {
          const float dl = sqrtf(dl2);
          const float eA = reachNV(cq, mulVM(-dir, tq.b));
          const float mq = cvxMargin<1>(cq);
          const float deltaS = dl2 - eA - (rad + mq) * dl;
          if ((deltaS > 0.0f && deltaS > lim * dl) || ca.dbg == 2) {
            keep = false;
          } else if (TOHC_BP_EXACT) {
            const ChildDesc dx = sc.child_desc[x];
            const Xf tx = xDyn ? ldXf(a.dynXf0 + ((size_t)set0 * dm.n_dyn_children + xSlot) * 12) : ldXf(sc.stat_xf + 12 * x);
            Cvx<1> cx;
            cvxLoad<1>(cx, dx.kind, mk(dx.dims[0], dx.dims[1], dx.dims[2]), sc.vx, sc.vy, sc.vz, sc.v4, dx.voff, dx.vcnt, sc.v8, dx.voff8);
            keep = !exactBeyond(dir, dl2, tq.o, eA, mq, cx, tx, lim);
          }
        }
      }
      const unsigned long long m = __ballot(keep);
      const unsigned int cnt = (unsigned int)__popcll(m);
      if (cnt > 0) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&a.cnt->sh[shard].n_pairs, cnt);
        base = __shfl(base, 0);
        if (keep) {
          const unsigned int pos = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
          if (pos < cp.seg_cap) {
            PairRec p;
            p.set = cs; p.q = q; p.x = x; p.pad = 0;
            a.pairs[(size_t)shard * cp.seg_cap + pos] = p;
          } else {
            a.cnt->overflow = 1;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    nq = 0;
  };

  for (int qi = 0; qi < ca.nQ; ++qi) {
    const int4 qinfo = reinterpret_cast<const int4 *>(ca.qchildren)[qi];
    const int q = qinfo.x, qs = qinfo.z;
    Aabb bq = qs >= 0 ? ldAabb(ca.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + qs) * 6) : ldAabb(sc.stat_aabb + 6 * q);
    if (CAST && qs >= 0) bq = aabbUnion(bq, ldAabb(ca.dynAabb1 + ((size_t)set1 * dm.n_dyn_children + qs) * 6));
    for (int ob = 0; ob < nO; ob += TOHC_BP_THREADS) {
      const int o = ob + lane;
      bool hit = false;
      unsigned int ent = (unsigned int)qi << 23;
      if (o < nStat) {           // pad entries hold inverted boxes: they never overlap
        const float *t = dm.stat_soa + o;
        const int P = dm.n_stat_pad;
        Aabb bx;
        bx.lo = mk(t[0], t[P], t[2 * P]);
        bx.hi = mk(t[3 * P], t[4 * P], t[5 * P]);
        hit = aabbOverlap(bq, bx);
        ent |= (unsigned int)dm.stat_children[o];
      } else if (o < nO) {
        hit = aabbOverlap(bq, ldAabb(ca.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + (o - nStat)) * 6));
        ent |= TOHC_BPU_DYN | (unsigned int)(o - nStat);
      }
      const unsigned long long m = __ballot(hit);
      if (hit) s_q[nq + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = ent;
      nq += (unsigned int)__popcll(m);
      if (nq > TOHC_BPU_QCAP - TOHC_BP_THREADS) drain();
    }
  }
  if (nq > 0) drain();
  if (lane == 0 && ncand > 0) atomicAdd(&a.cnt->sh[shard].n_cand, ncand);
}

// -----------------------------------------------------------------------------------------------------------------
// k_sweep: ContinuousCheckTrajectory (bullet_collision_checker.cpp:683-753).  ONE WAVEFRONT per (trajectory segment,
// swept geometry) = one convexSweepTest (btCollisionWorld.cpp:994-1056) with the ClosestConvexResultCallback:
//   * cull: the box that holds the geometry at every pose of the interpolated motion (its box at the start rotation,
//     grown by |angVel| * getAngularMotionDisc, swept along the origin's straight path) + `grow` (the 1 mm hit radius
//     of calcTimeOfImpact, the jitter the advancement applies to the resting side, a guard band) against the cluster
//     table / the moving children, like k_broadphase.  Bullet's own ray-vs-fattened-box cull is history dependent; both
//     are conservative (while 3.001 d >= 1 mm), so the surviving hit is the same;
//   * survivors are taken four at a time, one group of 16 lanes (a DPP row) each: hull vertices in registers,
//     conservative advancement with the full pair detector per step (timeOfImpact<16>), EPA pools in LDS;
//   * the closest hit wins (lowest fraction; ties -> lowest scene-wide child number, where the reference takes the
//     first in DBVT order), written as one record: idA/childA the swept geometry, idB/childB what it hits, time =
//     fraction, ptA = ptB = hit point, n = hit normal, dist = 0 (the arguments of the Collision the reference builds).
// Filters: KinBodyFilter objects only, no allowed-pair matrix (ccc.m_collisionFilterMask, :688; ConvexResultCallback::
// needsCollision does not consult it).
// -----------------------------------------------------------------------------------------------------------------
struct SweepArgs {
  NarrowArgs a;
  const int *qchildren;
  int nQ;
  const float *dynAabb0;
  float grow;
};
__global__ void __launch_bounds__(64) k_sweep(SweepArgs sa) {
  const NarrowArgs &a = sa.a;
  const DevScene &sc = a.sc;
  const DynMap &dm = a.dm;
  const CallParams &cp = a.cp;
  __shared__ EpaMem s_mem[4];
  __shared__ int s_q[TOHC_BP_QCAP];
  const int lane = threadIdx.x, grp = lane >> 4;
  const int qi = blockIdx.x % sa.nQ;
  const int cs = cp.unit_base + blockIdx.x / sa.nQ;
  const int q = sa.qchildren[4 * qi];
  int set0, set1;
  castSets(cp, cs, set0, set1);
  const ChildDesc dq = sc.child_desc[q];
  Cvx<16> cq;
  Xf tf0;
  loadSide<16>(a, q, dq, set0, set1, false, cq, tf0);
  const Xf tf1 = childXf(a, a.dynXf1, set1, q);
  const int shq = sc.child_shape[q];
  const V3 qmin = mk(sc.shape_lmin[3 * shq], sc.shape_lmin[3 * shq + 1], sc.shape_lmin[3 * shq + 2]);
  const V3 qmax = mk(sc.shape_lmax[3 * shq], sc.shape_lmax[3 * shq + 1], sc.shape_lmax[3 * shq + 2]);
  const float discA = angularMotionDisc(dq.kind, mk(dq.dims[0], dq.dims[1], dq.dims[2]), qmin, qmax, cp.d);
  Aabb bq;
  {
    V3 lin, ang;
    calcVelocity(tf0, tf1, lin, ang);
    Xf r0 = tf0;
    r0.o = mk(0, 0, 0);
    const Aabb b0 = shapeAabb(qmin, qmax, r0, 0.0f);
    const float g = len(ang) * discA * 1.001f + sa.grow;
    bq.lo = mk(fminf(tf0.o.x, tf1.o.x) + b0.lo.x - g, fminf(tf0.o.y, tf1.o.y) + b0.lo.y - g, fminf(tf0.o.z, tf1.o.z) + b0.lo.z - g);
    bq.hi = mk(fmaxf(tf0.o.x, tf1.o.x) + b0.hi.x + g, fmaxf(tf0.o.y, tf1.o.y) + b0.hi.y + g, fmaxf(tf0.o.z, tf1.o.z) + b0.hi.z + g);
  }
  unsigned int nq = 0;   // wave-uniform
  unsigned long long bestKey = ~0ull;
  SweepOut best;
  best.fraction = 0; best.normal = mk(0, 0, 0); best.point = mk(0, 0, 0);

  auto drain = [&]() {
    __syncthreads();   // one wavefront per block: orders the queue writes before the reads below
    for (unsigned int k = grp; k < nq; k += 4) {
      const int x = s_q[k];
      const ChildDesc dx = sc.child_desc[x];
      if (!(dx.group & TOHC_GROUP_KINBODY)) continue;   // ccc.m_collisionFilterMask = KinBodyFilter (bcc:688)
      Cvx<16> cx;
      Xf tx;
      loadSide<16>(a, x, dx, set0, set1, false, cx, tx);
      const int shx = sc.child_shape[x];
      const float discB = angularMotionDisc(dx.kind, mk(dx.dims[0], dx.dims[1], dx.dims[2]), mk(sc.shape_lmin[3 * shx], sc.shape_lmin[3 * shx + 1], sc.shape_lmin[3 * shx + 2]),
                                            mk(sc.shape_lmax[3 * shx], sc.shape_lmax[3 * shx + 1], sc.shape_lmax[3 * shx + 2]), cp.d);
      SweepOut so;
      if (timeOfImpact<16>(&s_mem[grp], cq, discA, tf0, tf1, cx, discB, tx, so)) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(so.fraction) << 32) | (unsigned int)x;
        if (key < bestKey) { bestKey = key; best = so; }
      }
    }
    __syncthreads();
    nq = 0;
  };
  auto push = [&](bool hit, int x) {
    const unsigned long long m = __ballot(hit);
    if (hit) s_q[nq + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = x;
    nq += (unsigned int)__popcll(m);
  };
  // The scan is a producer with ONE consumer call site: fill the queue, drain, repeat.  (With drain() called from
  // several places the compiler keeps it as a function, and everything it captures -- both hulls, the poses, the
  // best hit -- then lives in scratch memory: 1.1 KB per lane, 330 MB of HBM writes per launch.)
  int stage = dm.n_cl > 0 ? 0 : 1;   // 0: clusters of the table, 1: moving children, 2: done
  int cb = -TOHC_BP_THREADS, base = 0;
  unsigned long long cm = 0;
  while (stage < 2 || nq > 0) {
    while (stage < 2 && nq <= TOHC_BP_QCAP - TOHC_BP_THREADS) {
      if (stage == 0) {
        if (cm == 0) {   // next block of 64 cluster boxes
          cb += TOHC_BP_THREADS;
          if (cb >= dm.n_cl) { stage = 1; continue; }
          const int c = cb + lane;
          bool over = false;
          if (c < dm.n_cl) {
            const float *t = dm.cl_soa + c;
            const int P = dm.n_cl_pad;
            Aabb bc;
            bc.lo = mk(t[0], t[P], t[2 * P]);
            bc.hi = mk(t[3 * P], t[4 * P], t[5 * P]);
            over = aabbOverlap(bq, bc);
          }
          cm = __ballot(over);
          continue;
        }
        const int cc = cb + (int)__builtin_ctzll(cm);
        cm &= cm - 1;
        const int i = cc * 64 + lane;
        const float *t = dm.stat_soa + i;
        const int P = dm.n_stat_pad;
        Aabb bx;
        bx.lo = mk(t[0], t[P], t[2 * P]);
        bx.hi = mk(t[3 * P], t[4 * P], t[5 * P]);
        const int chx = dm.stat_children[i];
        push(aabbOverlap(bq, bx), chx);
      } else {
        if (base >= dm.n_dyn_children) { stage = 2; continue; }
        const int i = base + lane;
        bool hit = false;
        int chx = 0;
        if (i < dm.n_dyn_children) {
          chx = dm.dyn_children[i];
          hit = aabbOverlap(bq, ldAabb(sa.dynAabb0 + ((size_t)set0 * dm.n_dyn_children + i) * 6));
        }
        push(hit, chx);
        base += TOHC_BP_THREADS;
      }
    }
    drain();
  }
  // the closest of the four groups' hits
  unsigned long long k = bestKey;
  for (int g = 0; g < 4; ++g) {
    const unsigned long long o = __shfl(bestKey, g * 16);
    k = o < k ? o : k;
  }
  if (k != ~0ull && k == bestKey && (lane & 15) == 0) {   // x is part of the key: exactly one group holds it
    const unsigned int slot = atomicAdd(a.n_records, 1u);
    if (slot < cp.max_records) {
      const int x = (int)(unsigned int)(k & 0xffffffffull);
      tohc_record r;
      r.problem = set0 / cp.T;
      r.t = set0 % cp.T;
      r.idA = sc.child_obj[q]; r.childA = sc.child_idx[q]; r.idB = sc.child_obj[x]; r.childB = sc.child_idx[x];
      r.cctype = TOHC_CC_NONE;
      r.time = best.fraction;
      r.dist = 0;
      r.n[0] = best.normal.x; r.n[1] = best.normal.y; r.n[2] = best.normal.z;
      r.ptA[0] = r.ptB[0] = best.point.x; r.ptA[1] = r.ptB[1] = best.point.y; r.ptA[2] = r.ptB[2] = best.point.z;
      r.ptB1[0] = r.ptB1[1] = r.ptB1[2] = 0;
      r.const0 = r.const1 = 0;
      a.records[slot] = r;
    } else {
      a.cnt->overflow = 1;
    }
  }
}

// Bullet's output filter [gjk:430-449] + the collector's distance filter (bullet_collision_checker.cpp:401)
__device__ __forceinline__ bool isContact(const NarrowArgs &a, const GjkState &st, float md2, PairOut &o) {
  gjkOutput(st, md2, o);
  return o.valid && !((double)o.distance > (double)a.cp.d);
}
__device__ __forceinline__ void writeHit(const NarrowArgs &a, int shard, unsigned int slot, unsigned int pairPos, const PairOut &o) {
  if (slot < a.cp.seg_cap) {
    HitRec h;
    h.pair = (int)pairPos;
    h.distance = o.distance;
    h.n[0] = o.normalOnB.x; h.n[1] = o.normalOnB.y; h.n[2] = o.normalOnB.z;
    h.pB[0] = o.pointOnB.x; h.pB[1] = o.pointOnB.y; h.pB[2] = o.pointOnB.z;
    a.hits[(size_t)shard * a.cp.seg_cap + slot] = h;
  } else {
    a.cnt->overflow = 1;
  }
}
// a finished pair evaluation: `o` goes to the hit list when it is a contact the collector keeps.  Call with every lane
// of the (possibly divergent) branch: the lane-per-pair kernels reserve their slots with one atomic per wavefront.
template <int G>
__device__ __forceinline__ void pushOut(const NarrowArgs &a, unsigned int pairPos, const PairOut &o) {
  const bool hit = o.valid && !((double)o.distance > (double)a.cp.d);
  const int shard = blockIdx.x % TOHC_NSHARD;
  if (G == 1) {
    const unsigned int slot = waveSlot(&a.cnt->sh[shard].n_hits, hit);
    if (hit) writeHit(a, shard, slot, pairPos, o);
  } else if (hit && Grp<G>::lane() == 0) {
    writeHit(a, shard, atomicAdd(&a.cnt->sh[shard].n_hits, 1u), pairPos, o);
  }
}
template <int G>
__device__ __forceinline__ void pushIfContact(const NarrowArgs &a, unsigned int pairPos, const GjkState &st, float md2) {
  PairOut o;
  const bool hit = isContact(a, st, md2, o);
  const int shard = blockIdx.x % TOHC_NSHARD;
  if (G == 1) {
    const unsigned int slot = waveSlot(&a.cnt->sh[shard].n_hits, hit);
    if (hit) writeHit(a, shard, slot, pairPos, o);
  } else if (hit && Grp<G>::lane() == 0) {
    writeHit(a, shard, atomicAdd(&a.cnt->sh[shard].n_hits, 1u), pairPos, o);
  }
}

// the end of a GJK run: the pair goes to the penetration queue (EPA) and / or becomes a hit
template <int G>
__device__ __forceinline__ void gjkDone(const NarrowArgs &a, unsigned int pos, GjkRun<G> &g, const Cvx<G> &cA, const Cvx<G> &cB, float md2p) {
  GjkState st;
  g.finish(st, cvxMargin<G>(cA), cvxMargin<G>(cB));
  const int shard = blockIdx.x % TOHC_NSHARD;
  unsigned int slot = 0;
  bool writeE = st.needEpa;
  if (G == 1) slot = waveSlot(&a.cnt->sh[shard].n_epa, st.needEpa);
  else if (st.needEpa && Grp<G>::lane() == 0) slot = atomicAdd(&a.cnt->sh[shard].n_epa, 1u);
  else writeE = false;
  if (writeE) {
    if (slot < a.cp.seg_cap) {
      EpaRec e;
      e.pair = (int)pos;
      e.isValid = st.isValid;
      e.distance = st.distance;
      e.n[0] = st.normalInB.x; e.n[1] = st.normalInB.y; e.n[2] = st.normalInB.z;
      e.pB[0] = st.pointOnB.x; e.pB[1] = st.pointOnB.y; e.pB[2] = st.pointOnB.z;
      a.epaq[(size_t)shard * a.cp.seg_cap + slot] = e;
    } else {
      a.cnt->overflow = 1;
    }
  }
  // (every lane that finished in this step takes part in the vote inside; pairs on their way to EPA are no contact yet)
  GjkState sh = st;
  if (st.needEpa) sh.isValid = false;
  pushIfContact<G>(a, pos, sh, md2p);
}
template <int G>
__device__ __forceinline__ void gjkSave(GjkCont &o, unsigned int pos, const GjkRun<G> &g) {
  GjkCont r;
  r.pos = pos; r.curIter = g.curIter; r.n = g.sx.n; r.cachedValid = g.sx.cachedValid ? 1 : 0;
  r.squaredDistance = g.squaredDistance;
  st3(r.sep, g.sep); st3(r.lastW, g.sx.lastW); st3(r.cP1, g.sx.cP1); st3(r.cP2, g.sx.cP2);
#pragma unroll
  for (int k = 0; k < 3; ++k) { st3(r.W[k], g.sx.W[k]); st3(r.P[k], g.sx.P[k]); st3(r.Q[k], g.sx.Q[k]); }
  o = r;
}
// (after g.init with the pair's poses)
__device__ __forceinline__ void gjkRestore(GjkRun<1> &g, const GjkCont &r) {
  g.curIter = r.curIter; g.squaredDistance = r.squaredDistance;
  g.sep = mk(r.sep[0], r.sep[1], r.sep[2]);
  g.sx.n = r.n; g.sx.cachedValid = r.cachedValid != 0; g.sx.needsUpdate = false;
  g.sx.lastW = mk(r.lastW[0], r.lastW[1], r.lastW[2]);
  g.sx.cP1 = mk(r.cP1[0], r.cP1[1], r.cP1[2]); g.sx.cP2 = mk(r.cP2[0], r.cP2[1], r.cP2[2]);
  g.sx.cV = g.sep;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g.sx.W[k] = mk(r.W[k][0], r.W[k][1], r.W[k][2]); g.sx.P[k] = mk(r.P[k][0], r.P[k][1], r.P[k][2]); g.sx.Q[k] = mk(r.Q[k][0], r.Q[k][1], r.Q[k][2]);
  }
  g.sx.W[3] = g.sx.P[3] = g.sx.Q[3] = mk(0, 0, 0);
}

// Pairs are handed out in runs of TOHC_RUN consecutive list entries per group: the broadphase writes the survivors of
// one (unit, query child) block contiguously, so within a run the query side (shape registers, pose, swept transform)
// is usually unchanged and only the other side is reloaded.
#define TOHC_RUN 8
template <int G, bool CAST, int BLOCK>
#ifndef TOHC_GJK_WAVES
#define TOHC_GJK_WAVES 2
#endif
__global__ void __launch_bounds__(BLOCK, TOHC_GJK_WAVES) k_gjk(NarrowArgs a) {
  const int groupsPerBlock = BLOCK / G;
  const unsigned int gid = blockIdx.x * groupsPerBlock + threadIdx.x / G;
  const unsigned int ngroups = gridDim.x * groupsPerBlock;
  QPrefix pre;
  qprefix(&a.cnt->sh[0].n_pairs, a.cp.seg_cap, pre);
  const unsigned int npairs = pre.start[TOHC_NSHARD];
  const int maxIter = a.cp.dbg_max_iter;
  // operands live in their GJK roles; the query child stays in its role register across pairs when it can
  Cvx<G> cA, cB;
  Xf tA, tB;
  GjkRun<G> g;
  int lastQ = -1, lastSet = -1;
  bool qInB = false;
  bool active = false;
  // short lists (after the pre-test they usually are): shorter runs, so that every group of the grid has work
  const unsigned int runLen = min((unsigned int)TOHC_RUN, max(1u, (npairs + ngroups - 1) / ngroups));
  unsigned int run = gid * runLen, p = run, runEnd = min(run + runLen, npairs);
  unsigned int pos = 0;
  unsigned int steps = 0;   // loop-body passes of this lane (groups: counted by their first lane)
  for (;;) {
    if (!active) {
      if (p >= runEnd) {  // this group's run is used up (or was empty): take the next one
        run += ngroups * runLen;
        if (run >= npairs) break;
        p = run;
        runEnd = min(run + runLen, npairs);
      }
      pos = qlocate(pre, a.cp.seg_cap, p);
      ++p;
      const PairRec pr = a.pairs[pos];
      const ChildDesc dx = a.sc.child_desc[pr.x];
      int set0, set1;
      castSets(a.cp, pr.set, set0, set1);
      // GJK roles: X compound -> A = X child, B = query child; else A = query child, B = X
      const bool xIsA = dx.compound != 0;
      const bool needQ = !(pr.q == lastQ && pr.set == lastSet && qInB == xIsA);
      if (xIsA) {
        loadSide<G>(a, pr.x, dx, set0, set1, false, cA, tA);
        if (needQ) loadSide<G>(a, pr.q, a.sc.child_desc[pr.q], set0, set1, CAST, cB, tB);
      } else {
        loadSide<G>(a, pr.x, dx, set0, set1, false, cB, tB);
        if (needQ) loadSide<G>(a, pr.q, a.sc.child_desc[pr.q], set0, set1, CAST, cA, tA);
      }
      lastQ = pr.q; lastSet = pr.set; qInB = xIsA;
      if (isSphereSphere<G>(cA, cB)) {   // discrete, both plain spheres: Bullet's analytic algorithm, no GJK
        PairOut o;
        sphereSphere(tA.o, shapeMargin<G>(cA), tB.o, shapeMargin<G>(cB), o);
        pushOut<G>(a, pos, o);
        continue;
      }
      if (CAST && xIsA && cA.kind == KIND_SPHERE && cB.kind == KIND_SPHERE &&
          castSphereChildCulled(tB.o, xfApply(tB, cB.t01.o), shapeMargin<G>(cB), tA.o, shapeMargin<G>(cA)))
        continue;   // the reference's child-box test removes this pair before GJK (see castSphereChildCulled)
      g.init(tA, tB);
      active = true;
    }
    if (Grp<G>::lane() == 0) ++steps;
    // (marginA + marginB + gContactBreakingThreshold)^2 [cca:370-373]: re-made from the operands where it is used, so
    // that it takes no register of its own across the loop (with zero margins it is the launch-wide constant)
    const float md2p = pairMaxDist2<G>(cA, cB, a.cp.breaking);
    if (g.step(cA, cB, tA, tB, md2p, maxIter)) {
      gjkDone<G>(a, pos, g, cA, cB, md2p);
      active = false;
    } else if (G == 1 && a.cp.gjk_split > 0 && g.curIter == a.cp.gjk_split) {
      // A wavefront lasts as long as its longest run, and the runs are uneven (most end within four passes, one in four
      // takes up to twice as many): the long ones go on in k_gjk_cont, packed into full wavefronts again.
      const int shard = blockIdx.x % TOHC_NSHARD;
      const unsigned int cap = a.cp.seg_cap / TOHC_CONT_DIV;
      const unsigned int slot = waveSlot(&a.cnt->sh[shard].n_cont, true);
      if (slot < cap) {   // (a full queue: the run simply stays)
        gjkSave(a.contq[(size_t)shard * cap + slot], pos, g);
        active = false;
      }
    }
  }
  waveAdd(&a.cnt->sh[blockIdx.x % TOHC_NSHARD].n_gjk_steps, steps);
}

// k_gjk_cont: the GJK runs k_gjk left unfinished, ONE LANE per run, from where they stopped (same operands, same state,
// same loop: the results are k_gjk's own).
#ifndef TOHC_CONT_GRID
#define TOHC_CONT_GRID 2048
#endif
template <bool CAST>
__global__ void __launch_bounds__(64, TOHC_GJK_WAVES) k_gjk_cont(NarrowArgs a) {
  const unsigned int cap = a.cp.seg_cap / TOHC_CONT_DIV;
  QPrefix pre;
  qprefix(&a.cnt->sh[0].n_cont, cap, pre);
  const unsigned int n = pre.start[TOHC_NSHARD];
  const int maxIter = a.cp.dbg_max_iter;
  unsigned int steps = 0;
  for (unsigned int i = blockIdx.x * 64u + threadIdx.x; i < n; i += gridDim.x * 64u) {
    const GjkCont rec = a.contq[qlocate(pre, cap, i)];
    const PairRec pr = a.pairs[rec.pos];
    const ChildDesc dx = a.sc.child_desc[pr.x];
    int set0, set1;
    castSets(a.cp, pr.set, set0, set1);
    const bool xIsA = dx.compound != 0;
    Cvx<1> cA, cB;
    Xf tA, tB;
    if (xIsA) {
      loadSide<1>(a, pr.x, dx, set0, set1, false, cA, tA);
      loadSide<1>(a, pr.q, a.sc.child_desc[pr.q], set0, set1, CAST, cB, tB);
    } else {
      loadSide<1>(a, pr.x, dx, set0, set1, false, cB, tB);
      loadSide<1>(a, pr.q, a.sc.child_desc[pr.q], set0, set1, CAST, cA, tA);
    }
    GjkRun<1> g;
    g.init(tA, tB);
    gjkRestore(g, rec);
    const float md2p = pairMaxDist2<1>(cA, cB, a.cp.breaking);
    for (;;) {
      ++steps;
      if (g.step(cA, cB, tA, tB, md2p, maxIter)) break;
    }
    gjkDone<1>(a, rec.pos, g, cA, cB, md2p);
  }
  waveAdd(&a.cnt->sh[blockIdx.x % TOHC_NSHARD].n_gjk_steps, steps);
}

// both operands of a queued pair in their GJK roles + the GJK result so far
template <int G>
__device__ __forceinline__ void loadQueued(const NarrowArgs &a, const EpaRec &e, bool cast, Cvx<G> &A, Cvx<G> &B, Xf &tA, Xf &tB, GjkState &st) {
  const PairRec pr = a.pairs[e.pair];
  const ChildDesc dq = a.sc.child_desc[pr.q], dx = a.sc.child_desc[pr.x];
  int set0, set1;
  castSets(a.cp, pr.set, set0, set1);
  if (dx.compound) {
    loadSide<G>(a, pr.x, dx, set0, set1, false, A, tA);
    loadSide<G>(a, pr.q, dq, set0, set1, cast, B, tB);
  } else {
    loadSide<G>(a, pr.q, dq, set0, set1, cast, A, tA);
    loadSide<G>(a, pr.x, dx, set0, set1, false, B, tB);
  }
  st.isValid = e.isValid != 0;
  st.needEpa = true;
  st.distance = e.distance;
  st.normalInB = mk(e.n[0], e.n[1], e.n[2]);
  st.pointOnB = mk(e.pB[0], e.pB[1], e.pB[2]);
  st.offset = (tA.o + tB.o) * 0.5f;
  st.iters = 0;
  st.mA = cvxMargin<G>(A); st.mB = cvxMargin<G>(B);
}

// -----------------------------------------------------------------------------------------------------------------
// The penetration tail.  Only ~10 % of the pairs k_gjk iterates on need calcPenDepth, and one pass (discrete or cast) of
// one linearization does not hold enough of them to fill the chip: launched per pass, k_pen kept 0.55 waves per SIMD
// resident and k_epa 1.2.  So the passes that are open when the tail is launched (engine: launchTail) share ONE launch
// of each tail kernel: TailArgs carries the NarrowArgs of every open pass ("head"), a block works for head
// blockIdx % n_heads (wave-uniform, so the head's arguments stay scalar); k_tail_epa's blocks, when their own head's
// queue is used up, take pairs from the other heads' queues.
// -----------------------------------------------------------------------------------------------------------------
#define TOHC_MAX_HEADS 2
struct TailArgs {
  int n_heads;
  unsigned int n_dist_blocks;  // k_tail_epa: its first n_dist_blocks blocks run the Distance fall-back (dispatched first: they start beside the first EPA runs), the rest EPA
  NarrowArgs h[TOHC_MAX_HEADS];
};

// k_tail_pen: first stage of calcPenDepth for the pairs k_gjk queued, a group of G lanes per pair (G = 4: the two hulls sit in
// the group's registers, 16 vertices a lane, so a support query is 16 dot products and two DPP steps instead of a
// scan through L2; G = 1: one lane per pair, hulls read from memory).  GJK2 needs only four support vertices, so its
// whole state fits a 140-byte LDS record per group.  Pairs whose origin lies inside the Minkowski difference go on to
// k_tail_epa with that state (PenRec); for the rest the Distance fall-back runs here and the pair is finished.
#ifndef TOHC_PEN_WAVES
#define TOHC_PEN_WAVES 2
#endif
#ifndef TOHC_PEN_CLASSES
#define TOHC_PEN_CLASSES 1   // 0: a pair goes to the shard of its block, whatever its GJK2 took (A/B measurements)
#endif
// NH: heads in this launch (1: a constant index, the arguments stay where the launch put them); CAST: 0 / 1 = every
// pair of the launch is discrete / cast (known at compile time when there is one head), 2 = ask the head
template <int G, int NH, int CAST>
__global__ void __launch_bounds__(64, TOHC_PEN_WAVES) k_tail_pen(TailArgs ta) {
  __shared__ G2Mem mem[64 / G];
  const int head = NH == 1 ? 0 : (int)(blockIdx.x % (unsigned int)NH);
  const NarrowArgs &a = ta.h[head];
  const unsigned int bid = blockIdx.x / (unsigned int)NH, nblocks = gridDim.x / (unsigned int)NH;
  const unsigned int gid = (bid * 64 + threadIdx.x) / G;
  const unsigned int nlanes = nblocks * 64 / G;
  const bool cast = CAST == 2 ? a.cp.cast != 0 : CAST == 1;
  QPrefix pre;
  qprefix(&a.cnt->sh[0].n_epa, a.cp.seg_cap, pre);
  const unsigned int nepa = pre.start[TOHC_NSHARD];
  float md2 = a.cp.breaking;
  md2 *= md2;
  const unsigned int cap2 = a.cp.pen_cap;
  unsigned int sups = 0;
  for (unsigned int i = gid; i < nepa; i += nlanes) {
    const unsigned int qpos = qlocate(pre, a.cp.seg_cap, i);
    const EpaRec e = a.epaq[qpos];
    Cvx<G> A, B;
    Xf tA, tB, lA, lB;
    GjkState st;
    loadQueued<G>(a, e, cast, A, B, tA, tB, st);
    md2 = pairMaxDist2<G>(A, B, a.cp.breaking);
    penLocal(tA, tB, st.offset, lA, lB);
    V3 tmpA = mk(0, 0, 0), tmpB = mk(0, 0, 0), sep = mk(0, 0, 0);
    bool isValid2 = false;
    Gjk2<G, G2Mem> g;
    // G == 1: only the first half of stage 1 here.  One pair in nine has the origin outside and needs Distance's GJK2 as
    // well -- another run as long as the first, which EVERY wavefront of 64 pairs would sit through; those pairs are
    // noted (distq) and finished by a few lane-per-pair blocks at the end of k_tail_epa's grid, beside the EPA runs.
    const bool toEpa = G == 1 ? penInside<G, G2Mem>(&mem[threadIdx.x / G], A, B, lA, lB, g)
                              : penStage1<G, G2Mem>(&mem[threadIdx.x / G], A, B, lA, lB, g, isValid2, sep, tmpA, tmpB);
    const int shard = blockIdx.x % TOHC_NSHARD;
    int pshard = shard;
    unsigned int slot = 0;
    bool writeP = toEpa;
    if (G == 1 && TOHC_PEN_CLASSES) {
      // k_tail_epa runs eight pairs of the queue in lock step, and a wavefront lasts as long as its longest run.  How long
      // an EPA run gets shows in how soon this GJK2 enclosed the origin: <= 3 passes of its loop -> 9.8 EPA iterations on
      // average (7 - 13), 4 -> 6.0, 5 -> 4.0, >= 6 -> 2.3 (config 4, host emulation; correlation -0.76).  The queue's
      // shards are read one after the other, so filing a pair under its class -- four shards per class, the block number
      // picks one -- puts like beside like, the long runs first: 14 % fewer lock-step iterations in the CPU model
      // (scripts/epa_lockstep_model.py).  One atomic per class present in the wavefront.
      const unsigned int cls = g.iters <= 3u ? 0u : g.iters == 4u ? 1u : g.iters == 5u ? 2u : 3u;
      pshard = (int)(cls * (TOHC_NSHARD / 4) + (blockIdx.x & (TOHC_NSHARD / 4 - 1)));
      unsigned long long todo = __ballot(toEpa);
      while (todo) {
        const int ps = __shfl(pshard, (int)__builtin_ctzll(todo));
        const bool mine = toEpa && pshard == ps;
        const unsigned int sl = waveSlot(&a.cnt->sh[ps].n_pen, mine);
        if (mine) slot = sl;
        todo &= ~__ballot(mine);
      }
      // (a class's four shards are full -- one class can hold most of the queue: the pair goes to its block's shard like
      // any pair did before, or to the next one with room; a shard's count may then pass its capacity, readers stop at
      // the capacity: qprefix)
      if (toEpa && slot >= cap2) {
        atomicAdd(&a.cnt->sh[shard].n_pen_refiled, 1u);
        for (int k = 0; k < TOHC_NSHARD && slot >= cap2; ++k) {   // the block's own shard first, then whichever still has room
          pshard = (shard + k) % TOHC_NSHARD;
          slot = atomicAdd(&a.cnt->sh[pshard].n_pen, 1u);
        }
      }
    } else if (G == 1) slot = waveSlot(&a.cnt->sh[shard].n_pen, toEpa);
    else if (toEpa && Grp<G>::lane() == 0) slot = atomicAdd(&a.cnt->sh[shard].n_pen, 1u);
    else writeP = false;
    if (writeP) {
      if (slot < cap2) {
        PenRec *r = &a.penq[(size_t)pshard * cap2 + slot];
        r->e = e;
        penSave<G, G2Mem>(r->h, g);
      } else {
        a.cnt->overflow = 1;
      }
    }
    if (G == 1) {
      const unsigned int ds = waveSlot(&a.cnt->n_dist, !toEpa);   // (one pair in nine, one atomic per wavefront: a single counter will do)
      if (!toEpa) {
        if (ds < a.cp.seg_cap * TOHC_NSHARD) a.distq[ds] = qpos;
        else a.cnt->overflow = 1;
      }
    } else {
      if (toEpa) st.isValid = false;   // no contact yet: k_tail_epa decides
      else gjkFinishTail(isValid2, sep, tmpA, tmpB, st);
      pushIfContact<G>(a, (unsigned int)e.pair, st, md2);
    }
    if (Grp<G>::lane() == 0) sups += g.nsup;
  }
  waveAdd(&a.cnt->sh[blockIdx.x % TOHC_NSHARD].n_pen_sups, sups);
}

// k_tail_epa: second stage (EPA proper) for the pairs k_tail_pen handed over, one group of G lanes per pair, pools in LDS.
// Every run starts on small pools (EpaMemS, 2.5 KB a group instead of 6.3 KB: eight groups per wavefront at the
// occupancy the registers allow); a run that would leave them is repeated from the hand-over on the reference's full
// pools, which lie over the block's small ones -- the groups of the wavefront that need it take turns once all small
// runs of the round have ended.
// CAST: as in k_tail_pen (0 / 1: every pair of the launch is discrete / cast, 2: ask the head)
template <int G, int BLOCK, int NH, int CAST = 2>
__global__ void __launch_bounds__(BLOCK, TOHC_EPA_WAVES) k_tail_epa(TailArgs ta) {
  static_assert(BLOCK <= 64, "the groups of a block take turns on one full pool: one wavefront per block");
  constexpr int groupsPerBlock = BLOCK / G;
  constexpr size_t ldsEpa = sizeof(EpaMemS) * groupsPerBlock > sizeof(EpaMem) ? sizeof(EpaMemS) * groupsPerBlock : sizeof(EpaMem);
  constexpr size_t ldsBytes = ldsEpa > sizeof(G2Mem) * BLOCK ? ldsEpa : sizeof(G2Mem) * BLOCK;
  __shared__ __align__(16) unsigned char lds[ldsBytes];
  if (blockIdx.x < ta.n_dist_blocks) {
    // The Distance fall-back of the pairs k_tail_pen found the origin outside for (distq), one lane per pair, on the few
    // blocks at the head of the grid: they run beside the EPA runs instead of inside every wavefront of k_tail_pen.
    G2Mem *gm = reinterpret_cast<G2Mem *>(lds) + threadIdx.x;
    const unsigned int b = blockIdx.x, nb = ta.n_dist_blocks;
    unsigned int dsups = 0;
    for (int hh = 0; hh < NH; ++hh) {
      const NarrowArgs &a = ta.h[hh];
      unsigned int n = a.cnt->n_dist;
      if (n > a.cp.seg_cap * TOHC_NSHARD) n = a.cp.seg_cap * TOHC_NSHARD;
      for (unsigned int i = b * BLOCK + threadIdx.x; i < n; i += nb * BLOCK) {
        const EpaRec e = a.epaq[a.distq[i]];
        Cvx<1> A, B;
        Xf tA, tB, lA, lB;
        GjkState st;
        loadQueued<1>(a, e, CAST == 2 ? a.cp.cast != 0 : CAST == 1, A, B, tA, tB, st);
        const float md2 = pairMaxDist2<1>(A, B, a.cp.breaking);
        penLocal(tA, tB, st.offset, lA, lB);
        V3 tmpA = mk(0, 0, 0), tmpB = mk(0, 0, 0), sep = mk(0, 0, 0);
        bool isValid2 = false;
        Gjk2<1, G2Mem> g;
        penDistance<1, G2Mem>(gm, A, B, lA, lB, g, isValid2, sep, tmpA, tmpB);
        gjkFinishTail(isValid2, sep, tmpA, tmpB, st);
        pushIfContact<1>(a, (unsigned int)e.pair, st, md2);
        dsups += g.nsup;
      }
    }
    unsigned int *dctr = &ta.h[0].cnt->sh[blockIdx.x % TOHC_NSHARD].n_pen_sups;
    if (BLOCK >= 64) waveAdd(dctr, dsups);
    else if (dsups) atomicAdd(dctr, dsups);
    return;
  }
  EpaMemS *ms = reinterpret_cast<EpaMemS *>(lds) + threadIdx.x / G;
  EpaMem *mb = reinterpret_cast<EpaMem *>(lds);
  unsigned int sups = 0, redone = 0;
#if defined(TOHC_EPA_TIMELINE)   // diagnostics build: how long the wavefronts of this kernel live and how long their rounds last
  const unsigned long long tl_start = wall_clock64();   // (100 MHz)
#endif
  for (int hh = 0; hh < NH; ++hh) {   // the block's own head first, then whatever the others still hold
    const int head = NH == 1 ? 0 : (int)((blockIdx.x + (unsigned int)hh) % (unsigned int)NH);
    const NarrowArgs &a = ta.h[head];
    const unsigned int cap2 = a.cp.pen_cap;
    QPrefix pre;
    qprefix(&a.cnt->sh[0].n_pen, cap2, pre);
    const unsigned int npen = pre.start[TOHC_NSHARD];
    // Pairs are handed out on demand, BLOCK / G at a time per wavefront (a block is one wavefront or less): EPA runs take
    // from 2 to 255 iterations, and with a fixed share per group the kernel lasted as long as its unluckiest group.
    for (;;) {
      unsigned int first = 0;
      if (threadIdx.x == 0) first = atomicAdd(&a.cnt->epa_next, (unsigned int)groupsPerBlock);
      first = __shfl(first, 0);
      if (first >= npen) break;
#if defined(TOHC_EPA_TIMELINE)
      const unsigned long long tl_r0 = wall_clock64();
#endif
      const unsigned int i = first + threadIdx.x / G;
      const bool mine = i < npen;
      const PenRec *r = &a.penq[qlocate(pre, cap2, mine ? i : first)];
      const EpaRec e = r->e;
      Cvx<G> A, B;
      Xf tA, tB, lA, lB;
      GjkState st;
      loadQueued<G>(a, e, CAST == 2 ? a.cp.cast != 0 : CAST == 1, A, B, tA, tB, st);
      const float md2 = pairMaxDist2<G>(A, B, a.cp.breaking);
      penLocal(tA, tB, st.offset, lA, lB);
      V3 tmpA = mk(0, 0, 0), tmpB = mk(0, 0, 0), sep = mk(0, 0, 0);
      bool isValid2 = false;
      bool ok = true;
      unsigned int nsup = 0;
      if (mine) {
        Gjk2<G, EpaMemS> g;
        penRestore<G>(ms, r->h, A, B, lA, lB, g);
#if defined(TOHC_EPA_PROF)
        const unsigned long long tw0 = wall_clock64(), tc0 = clock64();
        EpaProf prof;
        ok = penStage2<G>(ms, lA, lB, g, isValid2, sep, tmpA, tmpB, &prof, a.cp.epa_small_v, a.cp.epa_small_f);
        if (Grp<G>::lane() == 0 && a.dbg) {
          atomicAdd(&a.dbg[0], 1ull);
          atomicAdd(&a.dbg[1], wall_clock64() - tw0);
          atomicAdd(&a.dbg[2], clock64() - tc0);
          atomicMax(&a.dbg[3], wall_clock64() - tw0);
          atomicAdd(&a.dbg[4], (unsigned long long)prof.nslow);
          atomicAdd(&a.dbg[5], (unsigned long long)prof.nfail);
          for (int k = 0; k < 8; ++k) { atomicAdd(&a.dbg[8 + k], prof.t[k]); atomicAdd(&a.dbg[16 + k], (unsigned long long)prof.n[k]); }
          atomicAdd(&a.dbg[32 + (prof.nsv < 63 ? prof.nsv : 63)], 1ull);
          atomicAdd(&a.dbg[96 + (prof.maxface < 127 ? prof.maxface : 127) / 2], 1ull);
        }
#else
        ok = penStage2<G>(ms, lA, lB, g, isValid2, sep, tmpA, tmpB, nullptr, a.cp.epa_small_v, a.cp.epa_small_f);
#endif
        nsup = g.nsup;
      }
      // (every lane of the wavefront is here again: no small pool is in use any more)
      unsigned long long redo = __ballot(!ok);
      while (redo) {
        const unsigned int grp = (unsigned int)__builtin_ctzll(redo) / G;
        if (threadIdx.x / G == grp) {
          Gjk2<G, EpaMem> g;
          penRestore<G>(mb, r->h, A, B, lA, lB, g);
          penStage2<G>(mb, lA, lB, g, isValid2, sep, tmpA, tmpB);
          nsup = g.nsup;
          ++redone;
        }
        redo &= ~((G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (grp * G));
      }
      if (mine) {
        gjkFinishTail(isValid2, sep, tmpA, tmpB, st);
        pushIfContact<G>(a, (unsigned int)e.pair, st, md2);
        if (Grp<G>::lane() == 0) sups += nsup;
      }
#if defined(TOHC_EPA_TIMELINE)
      if (threadIdx.x == 0 && a.dbg) {
        const unsigned long long dur = (wall_clock64() - tl_r0) / 1000ull;            // 10 us bins
        atomicAdd(&a.dbg[192 + (dur < 31 ? dur : 31)], 1ull);
        const unsigned long long at = (tl_r0 - tl_start) / 2500ull;                   // when the round began in the wavefront's life, 25 us bins
        atomicAdd(&a.dbg[224 + (at < 31 ? at : 31)], wall_clock64() - tl_r0);         // time spent in rounds begun in that bin
      }
#endif
    }
  }
#if defined(TOHC_EPA_TIMELINE)
  if (threadIdx.x == 0 && ta.h[0].dbg) {
    const unsigned long long life = (wall_clock64() - tl_start) / 2500ull;            // 25 us bins
    atomicAdd(&ta.h[0].dbg[160 + (life < 31 ? life : 31)], 1ull);
  }
#endif
  // (work counters only: booked on the block's own head)
  CounterShard *cs = &ta.h[NH == 1 ? 0 : blockIdx.x % (unsigned int)NH].cnt->sh[blockIdx.x % TOHC_NSHARD];
  if (Grp<G>::lane() != 0) redone = 0;
  if (BLOCK >= 64) { waveAdd(&cs->n_epa_sups, sups); waveAdd(&cs->n_epa_redone, redone); }
  else { if (sups) atomicAdd(&cs->n_epa_sups, sups); if (redone) atomicAdd(&cs->n_epa_redone, redone); }
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

// The same linearization with one lane per JOINT of the link's chain instead of one lane per column: a lane loads its
// joint's record and pose (independent loads, one round trip for the whole chain instead of a pointer chase per
// column) and, if the joint moves an optimisation column, leaves n . J[column] in the group's LDS row.  Values equal
// linearizeAt's: with at most one joint per column (DevRobot.fast_jac) a column's J is 0 + mult*col, exactly.
// row: [2][K] doubles of the group (A's then B's contributions).
template <int G>
__device__ __forceinline__ void jacRow(const DevRobot &rb, const double *linkQT, const double *linkZ, int4 jl, D3 n, D3 p, double *row) {
  const int gl = Grp<G>::lane();
  for (int c = jl.y + gl; c < jl.z; c += G) {
    const JacJoint jj = rb.jac_joint[c];
    const double *tq = linkQT + 7 * jj.link, *zq = linkZ + 3 * jj.link;
    const D3 z = mkd(zq[0], zq[1], zq[2]);   // = qrot(link pose, axis), bit for bit: k_fk does that product
    D3 col = jj.revolute ? cross(z, p - mkd(tq[4], tq[5], tq[6])) : z;
    D3 J = mkd(0, 0, 0);
    J.x += jj.mult * col.x; J.y += jj.mult * col.y; J.z += jj.mult * col.z;
    row[jj.col] = n.x * J.x + n.y * J.y + n.z * J.z;
  }
}
template <int G>
__device__ __forceinline__ double linearizeFast(const NarrowArgs &a, int set, int4 ja, int4 jb, D3 n, D3 ptA, D3 ptB, double dist, double *gradOut,
                                                double *row) {
  const DevRobot &rb = a.rb;
  const int gl = Grp<G>::lane();
  const int la = ja.x, lb = jb.x;
  const double *linkQT = a.linkQT + (size_t)set * rb.n_links * 7;
  const double *linkZ = a.linkZ + (size_t)set * rb.n_links * 3;
  const double *q = a.dofs + (size_t)set * rb.K;
  double *rowA = row, *rowB = row + rb.K;
  for (int k = gl; k < rb.K; k += G) {
    double va = 0, vb = 0;
    if (k >= rb.n_joint_inds) {  // affine columns: no chain involved
      if (la >= 0) { D3 J = jacColumn(rb, linkQT, q, la, ptA, k); va = n.x * J.x + n.y * J.y + n.z * J.z; }
      if (lb >= 0) { D3 J = jacColumn(rb, linkQT, q, lb, ptB, k); vb = n.x * J.x + n.y * J.y + n.z * J.z; }
    }
    rowA[k] = va; rowB[k] = vb;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the group's lanes share one wavefront: LDS ops are in order
  if (la >= 0) jacRow<G>(rb, linkQT, linkZ, ja, n, ptA, rowA);
  if (lb >= 0) jacRow<G>(rb, linkQT, linkZ, jb, n, ptB, rowB);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int k = gl; k < rb.K; k += G) {
    double g = 0;
    if (la >= 0) g += 1.0 * rowA[k];
    if (lb >= 0) g += -1.0 * rowB[k];
    gradOut[k] = g;
    rowA[k] = g;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double gq = 0;
  for (int k = 0; k < rb.K; ++k) gq += rowA[k] * q[k];   // column order, like linearizeAt
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return dist - gq;
}

// k_finish: btBridgedManifoldResult::addContactPoint (btCollisionWorld.cpp:1068-1086), CastCollisionCollector
// (bullet_collision_checker.cpp:894-972), record + gradient rows.
// CAST: the pass is a cast pass (a.cp.cast) -- a kernel of its own, so that the discrete pass's carries neither the
// classification's code nor its registers (the query hull's vertices at two poses).
#ifndef TOHC_FINISH_WAVES
#define TOHC_FINISH_WAVES 2   // no spills (4: 128 VGPRs, 73 spilled; measured 2: 0.205, 3: 0.26, 4: 0.22, 5: 0.31 ms per launch)
#endif
#ifndef TOHC_FINISH_WAVES_DISCRETE
#define TOHC_FINISH_WAVES_DISCRETE 4   // 104 VGPRs, no spills: four waves per SIMD (k_finish per pass, config 4: 0.230 -> 0.198 ms on average over the two passes)
#endif
template <int G, bool CAST>
__global__ void __launch_bounds__(256, CAST ? TOHC_FINISH_WAVES : TOHC_FINISH_WAVES_DISCRETE) k_finish(NarrowArgs a) {
  extern __shared__ double s_rows[];   // [256 / G][2][K]: the groups' gradient rows (launch: finishLds)
  double *row = s_rows + (size_t)(threadIdx.x / G) * 2 * a.rb.K;
  const bool fast = a.rb.fast_jac != 0;
  const int groupsPerBlock = 256 / G;
  const unsigned int gid = blockIdx.x * groupsPerBlock + threadIdx.x / G;
  const unsigned int ngroups = gridDim.x * groupsPerBlock;
  QPrefix pre;
  qprefix(&a.cnt->sh[0].n_hits, a.cp.seg_cap, pre);
  const unsigned int nhits = pre.start[TOHC_NSHARD];
  const unsigned int base = *a.rec_base;
  const int gl = Grp<G>::lane();
  for (unsigned int hIdx = gid; hIdx < nhits; hIdx += ngroups) {
    const unsigned int slot = base + hIdx;
    if (slot >= a.cp.max_records) {
      if (gl == 0) a.cnt->overflow = 1;
      continue;
    }
    const HitRec h = a.hits[qlocate(pre, a.cp.seg_cap, hIdx)];
    const PairRec pr = a.pairs[h.pair];
    // both children's descriptors and Jacobian entries in one round of loads
    const ChildDesc dq = a.sc.child_desc[pr.q];
    const ChildDesc *dxp = a.sc.child_desc + pr.x;
    const bool wantJ = a.cp.want_jac && a.linkQT;
    const int4 none = make_int4(-1, 0, 0, 0);
    const int4 jq = wantJ ? a.rb.child_jac[pr.q] : none, jx = wantJ ? a.rb.child_jac[pr.x] : none;
    PairIds c;
    castSets(a.cp, pr.set, c.set0, c.set1);
    c.Q = dq.obj; c.qc = dq.idx; c.X = dxp->obj; c.xc = dxp->idx;
    c.xIsA = dxp->compound != 0;
    int4 jA = c.xIsA ? jx : jq, jB = c.xIsA ? jq : jx;
    V3 nrm = mk(h.n[0], h.n[1], h.n[2]);
    V3 ptB = mk(h.pB[0], h.pB[1], h.pB[2]);
    V3 ptA = ptB + nrm * h.distance;
    int idA = c.xIsA ? c.X : c.Q, chA = c.xIsA ? c.xc : c.qc;
    int idB = c.xIsA ? c.Q : c.X, chB = c.xIsA ? c.qc : c.xc;
    int cctype = TOHC_CC_NONE;
    float time = 0;
    V3 ptB1 = mk(0, 0, 0);
    if (CAST) {
      const bool castFirst = !c.xIsA;
      Cvx<G> cq;
      cvxLoad<G>(cq, dq.kind, mk(dq.dims[0], dq.dims[1], dq.dims[2]), a.sc.vx, a.sc.vy, a.sc.vz, a.sc.v4, dq.voff, dq.vcnt, a.sc.v8, dq.voff8);
      Xf tf0 = childXf(a, a.dynXf0, c.set0, pr.q);
      Xf tq1 = childXf(a, a.dynXf1, c.set1, pr.q);
      Xf t01 = xfInverseTimes(tf0, tq1);
      Xf tw1 = xfMul(tf0, t01);  // m_cow->getWorldTransform() * shape->m_t01, bullet_collision_checker.cpp:904
      CastOut co;
      castClassify<G>(cq, tf0, tw1, nrm, ptA, ptB, castFirst, co);
      if (castFirst) {  // swap so that the swept link is B (bullet_collision_checker.cpp:913-917)
        V3 t = ptA; ptA = ptB; ptB = t;
        nrm = nrm * -1.0f;
        int ti = idA; idA = idB; idB = ti;
        ti = chA; chA = chB; chB = ti;
        const int4 tj = jA; jA = jB; jB = tj;
      }
      cctype = co.cctype;
      time = co.time;
      if (co.between) { ptB = co.ptB; ptB1 = co.ptB1; }
    }
    double c0 = 0, c1 = 0;
    if (wantJ) {
      const int la = jA.x, lb = jB.x;
      D3 n = mkd(nrm.x, nrm.y, nrm.z), pA = mkd(ptA.x, ptA.y, ptA.z), pB = mkd(ptB.x, ptB.y, ptB.z), pB1 = mkd(ptB1.x, ptB1.y, ptB1.z);
      double *g = a.grads + (size_t)slot * 2 * a.rb.K;
      c0 = fast ? linearizeFast<G>(a, c.set0, jA, jB, n, pA, pB, (double)h.distance, g, row)
                : linearizeAt<G>(a, c.set0, la, lb, n, pA, pB, (double)h.distance, g);
      if (CAST) {
        const D3 pBc = (cctype == TOHC_CC_BETWEEN) ? pB1 : pB;
        c1 = fast ? linearizeFast<G>(a, c.set1, jA, jB, n, pA, pBc, (double)h.distance, g + a.rb.K, row)
                  : linearizeAt<G>(a, c.set1, la, lb, n, pA, pBc, (double)h.distance, g + a.rb.K);
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
      r.dist = h.distance;
      r.n[0] = nrm.x; r.n[1] = nrm.y; r.n[2] = nrm.z;
      r.ptA[0] = ptA.x; r.ptA[1] = ptA.y; r.ptA[2] = ptA.z;
      r.ptB[0] = ptB.x; r.ptB[1] = ptB.y; r.ptB[2] = ptB.z;
      r.ptB1[0] = ptB1.x; r.ptB1[1] = ptB1.y; r.ptB1[2] = ptB1.z;
      r.const0 = c0; r.const1 = c1;
      a.records[slot] = r;
    }
  }
}

// between k_tail_epa and the k_finish launches: the hit counts of the open passes are final, so their record ranges are
// too.  Ranges are reserved in enqueue order (the heads are in that order; tails run one after another on one stream);
// the passes' k_finish kernels then run independently, side by side.
__global__ void k_tail_range(TailArgs ta) {
  for (int h = 0; h < ta.n_heads; ++h) {
    const NarrowArgs &a = ta.h[h];
    QPrefix pre;
    qprefix(&a.cnt->sh[0].n_hits, a.cp.seg_cap, pre);
    *const_cast<unsigned int *>(a.rec_base) = *a.n_records;
    *a.n_records += pre.start[TOHC_NSHARD];
  }
}

}  // namespace tohc
