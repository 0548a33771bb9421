// engine.hip -- libtrajopt_hipcc.so: the C-ABI of include/trajopt_hipcc.h and the gfx950 kernels behind it.
//
// Pipeline of one batch call (all on the engine's own stream, nothing allocates after warm-up):
//   per pass:  k_fk -> (k_cast_boxes) -> k_broadphase -> k_exact -> k_gjk -> k_gjk_cont    (a pass = one enqueue: discrete or cast;
//              tohc_enqueue_linearization: one k_fk and one k_broadphase_both for its two passes)
//   per tail:  k_tail_pen -> k_tail_epa -> k_tail_range            (ONE launch each for all passes open at that time)
//   per pass:  k_finish                                            (kernels and layouts: tohc_kernels.h)
// Static obstacles are transformed once at upload (the reference re-syncs every object on every call,
// bullet_collision_checker.cpp:629-655).  See DESIGN.md for layouts and rooflines.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/trajopt_hipcc.h"
#include "tohc_sort.h"
#include "tohc_kernels.h"
#include "tohc_hinge.h"

using namespace tohc;

// -----------------------------------------------------------------------------------------------------------------
// host side
// -----------------------------------------------------------------------------------------------------------------
template <class T>
struct DBuf {  // grow-only device buffer; released with its owner (the engine's device must be current then)
  T *p = nullptr;
  size_t cap = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
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
  HBuf() = default;
  HBuf(const HBuf &) = delete;
  HBuf &operator=(const HBuf &) = delete;
  ~HBuf() { release(); }
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
  std::vector<float> h_shape_dims, h_shape_lmin, h_shape_lmax, h_child_localT, h_shape_sph;
  std::vector<int> h_obj_off, h_obj_cnt, h_obj_compound, h_obj_group;
  std::vector<double> h_obj_static_T;
  std::vector<unsigned char> h_allowed;
  DBuf<int> d_shape_kind, d_shape_voff, d_shape_vcnt, d_child_shape, d_child_obj, d_child_idx, d_obj_off, d_obj_cnt, d_obj_compound, d_obj_group;
  DBuf<float> d_shape_dims, d_shape_lmin, d_shape_lmax, d_vx, d_vy, d_vz, d_child_localT, d_stat_xf, d_stat_aabb;
  DBuf<F4> d_v4;
  DBuf<float> d_v8;              // blocked hull vertices (tohc_narrow.h, Cvx<1>)
  std::vector<int> h_shape_voff8;
  DBuf<unsigned char> d_allowed;
  DBuf<ChildDesc> d_child_desc;
  DBuf<CullSph> d_stat_sph;
  float d = 0.05f, breaking = 0;

  // robot
  int n_links = 0, K = 0, n_joint_inds = 0, affine = 0, n_dof = 0;
  std::vector<int> h_parent, h_jtype, h_dof, h_dof_col, h_link_obj, h_link_dyn, h_obj_jaclink;
  double base_T[7], rot_axis[3];
  DBuf<int> d_parent, d_jtype, d_dof, d_dof_col, d_link_dyn, d_obj_jaclink, d_link_obj, d_chain_off, d_chain;
  std::vector<int> h_chain_off, h_chain;
  DBuf<int4> d_child_jac;          // k_finish's flattened Jacobian tables (DevRobot)
  DBuf<JacJoint> d_jac_joint;
  bool shared_columns = false;
  DBuf<double> d_T_origin, d_axis, d_mult, d_offs, d_q_robot;
  // dynamic maps: robot mode and all-objects ("poses") mode
  std::vector<int> h_obj_dynR, h_child_dynR, h_dyn_childrenR, h_qchildrenR;
  std::vector<int> h_obj_dynP, h_child_dynP, h_dyn_childrenP;
  std::vector<int> h_qinfoR, h_qinfoP;   // what d_qchildrenR / d_qchildrenP hold: 4 ints per query child (queryInfo)
  DBuf<int> d_obj_dynR, d_child_dynR, d_dyn_childrenR, d_qchildrenR, d_obj_dynP, d_child_dynP, d_dyn_childrenP, d_qchildrenP;
  int n_dyn_objR = 0, n_dyn_childrenR = 0;
  // children that never move in each mode, with their cull boxes as an SoA table [6][n_stat_pad]
  std::vector<int> h_stat_childrenR, h_stat_childrenP;
  DBuf<int> d_stat_childrenR, d_stat_childrenP;
  DBuf<float> d_stat_soaR, d_stat_soaP, d_cl_soaR;
  int n_statR = 0, n_stat_padR = 0, n_clR = 0, n_cl_padR = 0;

  // per-call buffers
  DBuf<double> d_dofs, d_linkQT, d_linkZ, d_objQT, d_objQT1, d_grads;
  DBuf<float> d_dynXf, d_dynAabb, d_dynXf1, d_dynAabb1;
  DBuf<CullSph> d_dynSph;   // [n_sets][moving children]: world bounding spheres at the first pose of each set (k_fk / k_dyn_children)
  DBuf<float> d_rays;              // RayCastCollision: [n_rays][8]
  DBuf<unsigned int> d_ray_closest;
  // Enqueues alternate between NLANE lanes, each with its own HIP stream and queues, so that the discrete and the
  // cast pass of one linearization (and consecutive enqueues in general) overlap on the GPU; only k_finish, which
  // appends to the shared record slab, waits for the previous enqueue's.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;       // recorded after the lane's latest k_finish
    hipEvent_t head_done = nullptr;  // recorded after the lane's latest k_gjk: the tail may start on that pass's queues
    DBuf<PairRec> pairs, cands;
    DBuf<float> castAabb;            // [sets][moving children][6]: second boxes of the lane's cast pass (k_cast_boxes)
    DBuf<EpaRec> epaq;
    DBuf<PenRec> penq;
    DBuf<unsigned int> distq;
    DBuf<GjkCont> contq;
    DBuf<HitRec> hits;
  };
  Lane lane[4];
  int n_lanes = 4;                   // TOHC_STREAMS=1..4
  int chunks = 1;                    // a large enqueue is cut into this many unit ranges, one per lane (TOHC_CHUNKS)
  hipEvent_t ev_fk = nullptr;        // the FK of the staged DOF vectors has finished (main stream)
  hipEvent_t ev_prep = nullptr, ev_fused = nullptr;   // tohc_enqueue_linearization: the first pass's lane is ready / the shared broadphase has finished
  int fuse_broadphase = 1;           // TOHC_FUSE_BP=0: tohc_enqueue_linearization launches the two passes' broadphases separately
  // Passes whose head (broadphase + GJK) is on its lane and whose tail (penetration, EPA, record ranges, k_finish) is
  // still to be launched: the tail kernels take ALL open passes in one launch each (launchTail), which is what fills
  // the chip -- one pass alone queues fewer penetration pairs than the GPU has wave slots.  A tail goes out when
  // max_open passes are open and, at the latest, when the host asks for results (sync / fetch) or touches the tables.
  struct OpenHead { int slot; int lane; bool empty; NarrowArgs a; };
  std::vector<OpenHead> open;
  // max_open = 1 (default): every pass gets a tail of its own on its own lane.  TOHC_MERGE_TAIL=1: two passes share a
  // tail.  Measured on config 4, 64 problems (gpurun_out/r02_d_*): the shared tail itself is shorter (1.47 vs 1.63 ms of
  // kernel time per step) but the step is longer (4.83 vs 4.55 ms), because a shared tail cannot start before BOTH heads
  // have finished and so no longer overlaps with the other pass's broadphase / GJK.
  int max_open = 1;
  // tohc_enqueue_linearization hands both passes' heads to the GPU before the first pass's tail (TOHC_HEADS_FIRST=0: pass by
  // pass).  The host needs ~0.1 ms to submit one pass's ten launches and event operations; for a single problem, whose
  // kernels last 10 - 100 us each, the second pass's head otherwise starts that much later than it could.
  int heads_first = 1;
  unsigned int pen_cap_limit = ~0u;   // TOHC_PEN_CAP=n: a shard of the penetration hand-over queue takes n pairs at most (tests of
                                      // k_tail_pen's fall-back from a full class shard to the block's shard)
  hipStream_t tail_stream = nullptr;
  hipEvent_t tail_ranged = nullptr;  // recorded after the latest k_tail_range
  bool have_ranged = false;
  bool fk_done = false;
  DBuf<unsigned long long> d_dbg;   // TOHC_EPA_PROF builds only
  DBuf<Counters> d_cnt;            // [1 + RING]: element 0 holds the slab's record counter (d_nrec), slot i is element 1 + i -- one D2H copy brings both
  DBuf<unsigned int> d_rec_base;   // [RING] first record of each enqueue (k_range)
  DBuf<tohc_record> d_records, d_sorted;
  DBuf<double> d_gsorted;
  DBuf<uint64_t> d_sort_keys;
  DBuf<uint32_t> d_sort_idx;
  DBuf<unsigned char> d_sort_tmp;
  int last_T = 1;
  bool children_in_object_order = true;   // obj_child_off is the running sum of obj_child_cnt (what the sort key assumes)
  int64_t last_sets = 0;
  bool host_sort = false;   // TOHC_HOST_SORT=1: order the slab on the host (the path taken anyway beyond the key's limits)
  HBuf<tohc_record> h_records, h_sorted;
  HBuf<double> h_grads, h_gsorted;
  HBuf<Counters> h_cnt;            // [1 + RING], the same layout
  size_t max_pairs = (size_t)1 << 22, max_records = (size_t)1 << 18;
  tohc_counters counters;
  // enqueue ring: every enqueue since the last sync has its own device counter slot and event set
  struct Pending { int slot; int64_t aabb_tests; int kind = -1; int units = 0; };   // kind: 0 discrete / 1 cast pass of the narrowphase pipeline, -1 anything else
  std::vector<Pending> pending;
  std::vector<hipEvent_t> events;   // RING * NEV
  struct { unsigned int *p = nullptr; } d_nrec, h_nrec;   // shared record counter: the first word of d_cnt / h_cnt
  HBuf<double> h_dofs;              // pinned staging of the DOF vectors
  int staged_B = 0, staged_T = 0;
  bool staged = false;
  bool last_robot_mode = true, last_want_jac = false;   // of the current slab (set by the enqueue that starts it; appends must agree)
  // What tohc_fetch may trust: 0 = no slab yet / a new one is being filled (pending enqueues), 1 = the last sync
  // covered every enqueue of the slab and found no overflow, -1 = that sync failed (capacity / HIP error): *h_nrec
  // may then exceed what d_records holds, and the slab is truncated -- fetch refuses it (TOHC_E_STATE).
  int slab_state = 0;
  bool sorted_valid = false;   // d_sorted / d_gsorted hold the current slab in canonical order
  // ---- hinge fetch (tohc_hinge.h): sort + assembly + copies on a stream of their own, so that they overlap the next
  // linearization's kernels; two host buffer sets alternate
  hipStream_t hinge_stream = nullptr;
  hipEvent_t ev_slab_read = nullptr;    // the assembly has read d_records / d_grads: a new slab may overwrite them
  hipEvent_t ev_hinge_small = nullptr;  // the per-row arrays and the column offsets are in host memory
  bool slab_read_pending = false, hinge_open = false;
  DBuf<uint64_t> d_hkeys;
  DBuf<uint32_t> d_hidx;
  DBuf<unsigned char> d_htmp;
  DBuf<double> d_hrhs, d_hdist, d_hwa, d_hwb, d_hval;
  DBuf<int32_t> d_hrowt, d_hcolidx;
  DBuf<unsigned int> d_hseg, d_hcnt;
  DBuf<long long> d_hrowoff, d_hcoloff;
  struct HingeHost {
    HBuf<double> rhs, dist, val;
    HBuf<int32_t> row_t, idx;
    HBuf<long long> row_off, col_off;
    size_t n = 0;
    int B = 0, T = 0, K = 0, gap = 0;
  };
  HingeHost hh[2];
  int hinge_cur = 0;
  // what the current slab holds, pass by pass (enqueue calls): the hinge assembly takes one discrete and / or one cast pass
  int slab_B = 0, slab_disc_passes = 0, slab_cast_passes = 0, slab_gap = 0;
  int gjk_group = 1;  // lanes per pair in k_gjk: 1 (default), 16 or 64 (TOHC_GJK_GROUP)
  // Small batches (TOHC_GJK_ADAPT=0: off).  One lane per pair is the mapping for a full chip: its hull scans are chains of L2
  // round trips that only many resident wavefronts hide.  One problem alone (config 4: ~7 K pairs per pass, 116 wavefronts)
  // leaves the chip empty and waits for every one of them; with 16 lanes per pair (hull vertices in registers, DPP
  // arg-max, no memory access inside the loop) the same list answers sooner -- measured on config 4, one problem:
  // 0.70 -> 0.64 ms host-delivered, two problems 0.534 -> 0.507 ms per step, three 0.60 -> 0.66 with 22 K pairs in the cast pass (gpurun_out/s4_*, s5_*).  How
  // many pairs a pass holds is only known on the device, so the choice follows the LAST pass of the same kind and the
  // same number of units (an SQP solve linearizes the same problem again and again; syncEngineImpl notes the counts);
  // without such a pass: one lane per pair.  Results do not depend on it (test_engine_variants_compute_the_same_slab).
  int gjk_adapt = 1;
  struct GjkHint { int units = -1; unsigned int pairs = 0; } gjk_hint[2];   // [cast]
  int pen_group = 1;   // lanes per pair in k_pen: 1, or 4 (hull vertices in registers; TOHC_PEN_GROUP=4: measured no faster, GJK2's LDS bookkeeping dominates)
  int epa_block = 64;  // threads per k_epa block (TOHC_EPA_BLOCK = 16 or 64)
  bool timing = true;  // per-stage timing events between the kernels (recEv)
  int unit_broadphase = 1024;   // k_broadphase_unit from this many units on (one wavefront per unit: fewer units leave the chip empty and
                                // a wavefront per (unit, query child) answers sooner); TOHC_UNIT_BP=n sets it, 0 = never
  unsigned int epa_small_v = TOHC_EPA_SMALL_V, epa_small_f = TOHC_EPA_SMALL_F;   // TOHC_EPA_SMALL=v,f shrinks k_tail_epa's first-attempt pools (tests of the repeat path)
  int gjk_split = 6;     // TOHC_GJK_SPLIT=n: k_gjk hands runs still going after n passes to k_gjk_cont (0: never; measured on config 4: 0: 2.555 ms per step, 3: 2.58, 5: 2.50, 6: 2.49, 7: 2.52, 8: 2.56)
  int defer_exact = 1;   // TOHC_DEFER_EXACT=0: the exact stage of the pre-test stays inside k_broadphase
  int pretest = 1;   // TOHC_PRETEST=0 turns the separating-axis pre-test off (A/B measurements)
  tohc_engine() { memset(&counters, 0, sizeof counters); }
};
static const int RING = 32;
// events per enqueue: [0] fk [1] (children) [2] broadphase [3] gjk [4] -- on the pass's lane; [5] tail (k_tail_pen,
// k_tail_epa, k_tail_range) [6] -- on the tail stream, real only for the last pass of a tail, a zero-length pair for
// the others; [7] k_finish [8] -- on the lane again
static const int NEV = 9;
// The per-stage timing events of an enqueue (NEV per slot: tohc_counters' cum_ms_*).  Each record is a marker packet between
// two kernels of a stream; with kernel timing off (tohc_set_kernel_timing / TOHC_TIMING=0) only the first and the last of a
// slot are recorded -- the span of the call, ms_last_call -- and the kernels of a stream follow each other directly.
static hipError_t recEv(tohc_engine *e, hipEvent_t *ev, int i, hipStream_t s);

static int fail(tohc_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

static int launchTail(tohc_engine *e);
static hipError_t recEv(tohc_engine *e, hipEvent_t *ev, int i, hipStream_t s) {
  if (!e->timing && i != 0 && i != NEV - 1) return hipSuccess;
  return hipEventRecord(ev[i], s);
}
// nothing may still be running on the lanes when tables they read are replaced
static hipError_t quiesce(tohc_engine *e) {
  if (launchTail(e) != TOHC_OK) return hipErrorUnknown;   // (e->err says what failed)
  for (auto &ln : e->lane)
    if (ln.stream) { hipError_t r = hipStreamSynchronize(ln.stream); if (r != hipSuccess) return r; }
  if (e->tail_stream) { hipError_t r = hipStreamSynchronize(e->tail_stream); if (r != hipSuccess) return r; }
  if (e->hinge_stream) { hipError_t r = hipStreamSynchronize(e->hinge_stream); if (r != hipSuccess) return r; }
  e->fk_done = false;
  e->gjk_hint[0] = e->gjk_hint[1] = tohc_engine::GjkHint();   // (a new scene, robot, distance or matrix: pair counts of the old one say nothing)
  return hipStreamSynchronize(e->stream);
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

// query children travel to the device as (child, its object, its slot among the moving children or -1, 0): one 16-byte
// load at the head of k_broadphase / k_sweep instead of a chain of three
static std::vector<int> queryInfo(const tohc_engine *e, const std::vector<int> &q, const std::vector<int> &child_dyn) {
  std::vector<int> out(4 * q.size());
  for (size_t i = 0; i < q.size(); ++i) {
    out[4 * i] = q[i]; out[4 * i + 1] = e->h_child_obj[q[i]]; out[4 * i + 2] = child_dyn[q[i]]; out[4 * i + 3] = 0;
  }
  return out;
}
static float safeMargin(float cur, float x, float y, float z) {  // btConvexInternalShape::setSafeMargin, btConvexInternalShape.h:70-85
  float mn = x < y ? (x < z ? x : z) : (y < z ? y : z);
  float safe = 0.1f * mn;
  return safe < cur ? safe : cur;
}

// -----------------------------------------------------------------------------------------------------------------
// Working memory of tohc_ray_cast_poses_noalloc: a fixed array in the library's device image (each GPU the library
// runs on holds its own copy, at its own address, until the process ends).  `scratch` itself is only the host-side
// handle of the symbol: host code asks for the device address once per device (scratchInit, from tohc_create, with that
// device current) and uses that pointer everywhere -- kernel arguments included.
//   layout: child poses [MAX_CHILDREN][12] fp32 | cull boxes [MAX_CHILDREN][6] fp32 | rays [MAX_RAYS][8] fp32
// -----------------------------------------------------------------------------------------------------------------
static const size_t SCRATCH_XF_FLOATS = (size_t)TOHC_NOALLOC_MAX_CHILDREN * 12, SCRATCH_BOX_FLOATS = (size_t)TOHC_NOALLOC_MAX_CHILDREN * 6,
                    SCRATCH_RAY_FLOATS = (size_t)TOHC_NOALLOC_MAX_RAYS * 8;
static __device__ __align__(256) float scratch[SCRATCH_XF_FLOATS + SCRATCH_BOX_FLOATS + SCRATCH_RAY_FLOATS];
static const int SCRATCH_MAX_DEVICES = 64;
static float *g_scratch_ptr[SCRATCH_MAX_DEVICES];   // device address of `scratch` on each device, null until scratchInit
static std::mutex g_scratch_mutex;
static hipError_t scratchInit(int device) {   // `device` is the current device
  if (device < 0 || device >= SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return hipSuccess;
  void *p = nullptr;
  const hipError_t r = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (r != hipSuccess) return r;
  if (!p) return hipErrorInvalidSymbol;
  g_scratch_ptr[device] = static_cast<float *>(p);
  return hipSuccess;
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
    tohc_destroy(e);
    return TOHC_E_HIP;
  }
  const char *g = getenv("TOHC_GROUP");
  if (g && atoi(g) == 64) e->group = 64;
  if (const char *pt = getenv("TOHC_PRETEST")) e->pretest = atoi(pt) != 0;
  if (const char *pt = getenv("TOHC_DEFER_EXACT")) e->defer_exact = atoi(pt) != 0;
  if (const char *pt = getenv("TOHC_GJK_SPLIT")) e->gjk_split = std::max(0, atoi(pt));
  if (const char *ns = getenv("TOHC_STREAMS")) e->n_lanes = std::min(std::max(atoi(ns), 1), 4);
  if (const char *nc = getenv("TOHC_CHUNKS")) e->chunks = std::min(std::max(atoi(nc), 1), 4);
  if (hipEventCreateWithFlags(&e->ev_fk, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->ev_prep, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fused, hipEventDisableTiming) != hipSuccess) { tohc_destroy(e); return TOHC_E_HIP; }
  if (const char *fb = getenv("TOHC_FUSE_BP")) e->fuse_broadphase = atoi(fb) != 0;
  for (auto &ln : e->lane)
    if (hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ln.done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ln.head_done, hipEventDisableTiming) != hipSuccess) {
      tohc_destroy(e);
      return TOHC_E_HIP;
    }
  if (hipStreamCreateWithFlags(&e->tail_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->tail_ranged, hipEventDisableTiming) != hipSuccess) {
    tohc_destroy(e);
    return TOHC_E_HIP;
  }
  // (a high-priority stream for the assembly was tried: the whole pipeline got slower, 8.5 vs 5.4 ms per batch)
  if (hipStreamCreateWithFlags(&e->hinge_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_slab_read, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_hinge_small, hipEventDisableTiming) != hipSuccess) {
    tohc_destroy(e);
    return TOHC_E_HIP;
  }
  if (const char *mt = getenv("TOHC_MERGE_TAIL")) e->max_open = atoi(mt) != 0 ? TOHC_MAX_HEADS : 1;
  if (const char *hf = getenv("TOHC_HEADS_FIRST")) e->heads_first = atoi(hf) != 0;
  if (const char *pc = getenv("TOHC_PEN_CAP")) e->pen_cap_limit = (unsigned int)std::max(atoi(pc), 1);
  if (const char *eb = getenv("TOHC_EPA_BLOCK")) e->epa_block = atoi(eb) == 16 ? 16 : 64;
  if (const char *ub = getenv("TOHC_UNIT_BP")) e->unit_broadphase = std::max(atoi(ub), 0);
  if (const char *tm = getenv("TOHC_TIMING")) e->timing = atoi(tm) != 0;
  if (const char *es = getenv("TOHC_EPA_SMALL")) {
    unsigned int v = 0, f = 0;
    if (sscanf(es, "%u,%u", &v, &f) == 2 && v >= 1 && f >= 4) { e->epa_small_v = std::min(v, (unsigned int)TOHC_EPA_SMALL_V); e->epa_small_f = std::min(f, (unsigned int)TOHC_EPA_SMALL_F); }
  }
  if (const char *pg = getenv("TOHC_PEN_GROUP")) e->pen_group = atoi(pg) == 4 ? 4 : 1;
  if (const char *hs = getenv("TOHC_HOST_SORT")) e->host_sort = atoi(hs) != 0;
  e->gjk_group = e->group == 64 ? 64 : 1;
  if (const char *gg = getenv("TOHC_GJK_GROUP")) { int v = atoi(gg); if (v == 1 || v == 16 || v == 64) e->gjk_group = v; }
  if (const char *ga = getenv("TOHC_GJK_ADAPT")) e->gjk_adapt = atoi(ga) != 0;
  e->breaking = (float)(2.001 * (double)e->d);
  e->events.resize((size_t)RING * NEV);
  for (auto &ev : e->events)
    if (hipEventCreate(&ev) != hipSuccess) { tohc_destroy(e); return TOHC_E_HIP; }
  if (e->d_cnt.reserve(1 + RING) != hipSuccess || e->d_rec_base.reserve(RING) != hipSuccess || e->h_cnt.reserve(1 + RING) != hipSuccess ||
      hipMemset(e->d_cnt.p, 0, sizeof(Counters)) != hipSuccess) { tohc_destroy(e); return TOHC_E_HIP; }
  e->d_nrec.p = reinterpret_cast<unsigned int *>(e->d_cnt.p);
  e->h_nrec.p = reinterpret_cast<unsigned int *>(e->h_cnt.p);
  *e->h_nrec.p = 0;   // a fetch before the first enqueue reads this word
  if (scratchInit(device) != hipSuccess) { tohc_destroy(e); return TOHC_E_HIP; }
  *out = e;
  return TOHC_OK;
}

void tohc_destroy(tohc_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (auto &ln : e->lane) {
    if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
    if (ln.done) (void)hipEventDestroy(ln.done);
    if (ln.head_done) (void)hipEventDestroy(ln.head_done);
  }
  if (e->tail_stream) { (void)hipStreamSynchronize(e->tail_stream); (void)hipStreamDestroy(e->tail_stream); }
  if (e->tail_ranged) (void)hipEventDestroy(e->tail_ranged);
  if (e->hinge_stream) { (void)hipStreamSynchronize(e->hinge_stream); (void)hipStreamDestroy(e->hinge_stream); }
  if (e->ev_slab_read) (void)hipEventDestroy(e->ev_slab_read);
  if (e->ev_hinge_small) (void)hipEventDestroy(e->ev_hinge_small);
  if (e->ev_fk) (void)hipEventDestroy(e->ev_fk);
  if (e->ev_prep) (void)hipEventDestroy(e->ev_prep);
  if (e->ev_fused) (void)hipEventDestroy(e->ev_fused);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  for (auto ev : e->events)
    if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;   // every DBuf / HBuf releases its memory in its destructor: the list cannot drift from the struct
}

const char *tohc_last_error(const tohc_engine *e) { return e ? e->err.c_str() : "null engine"; }

static int refreshStatic(tohc_engine *e) {
  // world transform + d-inflated box of every child as if its object were static (used when it is not dynamic)
  std::vector<float> xf((size_t)e->n_children * 12), bb((size_t)e->n_children * 6);
  for (int c = 0; c < e->n_children; ++c) {
    int obj = e->h_child_obj[c];
    Xf T = xfFrom7(&e->h_obj_static_T[7 * obj]);
    if (e->h_obj_compound[obj]) T = xfMul(T, ldXf(&e->h_child_localT[12 * c]));
    stXf(&xf[12 * c], T);
    int sh = e->h_child_shape[c];
    Aabb b = shapeAabb(mk(e->h_shape_lmin[3 * sh], e->h_shape_lmin[3 * sh + 1], e->h_shape_lmin[3 * sh + 2]),
                       mk(e->h_shape_lmax[3 * sh], e->h_shape_lmax[3 * sh + 1], e->h_shape_lmax[3 * sh + 2]), T, e->d);
    float *o = &bb[6 * c];
    o[0] = b.lo.x; o[1] = b.lo.y; o[2] = b.lo.z; o[3] = b.hi.x; o[4] = b.hi.y; o[5] = b.hi.z;
  }
  HIPCHK(upload(e->d_stat_xf, xf, e->stream));
  HIPCHK(upload(e->d_stat_aabb, bb, e->stream));
  {
    std::vector<ChildDesc> cd(e->n_children);
    for (int c = 0; c < e->n_children; ++c) {
      ChildDesc &d = cd[c];
      memset(&d, 0, sizeof d);
      const int sh = e->h_child_shape[c], obj = e->h_child_obj[c];
      d.obj = obj; d.idx = e->h_child_idx[c]; d.kind = e->h_shape_kind[sh]; d.compound = e->h_obj_compound[obj];
      d.voff = e->h_shape_voff[sh]; d.vcnt = e->h_shape_vcnt[sh]; d.voff8 = e->h_shape_voff8[sh];
      d.dynR = e->h_child_dynR.empty() ? -1 : e->h_child_dynR[c];
      d.group = e->h_obj_group[obj];
      for (int k = 0; k < 3; ++k) d.dims[k] = e->h_shape_dims[3 * sh + k];
      for (int k = 0; k < 4; ++k) d.sph[k] = e->h_shape_sph[4 * sh + k];
    }
    HIPCHK(upload(e->d_child_desc, cd, e->stream));
    // the sphere stage of the pre-test for children that never move: world sphere + filter words in one 32-byte record
    std::vector<CullSph> cs(e->n_children);
    for (int c = 0; c < e->n_children; ++c) {
      CullSph &o = cs[c];
      memset(&o, 0, sizeof o);
      const V3 w = xfApply(ldXf(&xf[12 * c]), mk(cd[c].sph[0], cd[c].sph[1], cd[c].sph[2]));
      o.c[0] = w.x; o.c[1] = w.y; o.c[2] = w.z; o.r = cd[c].sph[3];
      o.obj = cd[c].obj; o.group = cd[c].group;
    }
    HIPCHK(upload(e->d_stat_sph, cs, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  // SoA cull-box table of the children that never move in robot mode (everything until a robot is uploaded), cut into
  // clusters of 64 consecutive entries with one box per cluster.  The order is that of the leaves of a k-d split: the
  // set is sorted along the longest axis of its box centres' bounds and cut so that the left part holds a multiple of 64
  // entries (half of them, rounded up to whole clusters), recursively; every cluster but the very last is a full leaf.
  // (Round 1 used 64 consecutive entries in Morton order of the centres: on config 4 a query box overlapped 2.8 of those
  // cluster boxes on average, 1.0 of these -- each overlap is a 64-lane round of box tests in k_broadphase / k_sweep.
  // Which children share a cluster never shows in the results.)
  std::vector<int> st = e->h_stat_childrenR;
  e->n_statR = (int)st.size();
  {
    std::vector<float> ctr(3 * st.size());
    for (size_t i = 0; i < st.size(); ++i)
      for (int k = 0; k < 3; ++k) {
        float m = 0.5f * (bb[6 * (size_t)st[i] + k] + bb[6 * (size_t)st[i] + 3 + k]);
        ctr[3 * i + k] = m == m ? m : 0.f;   // (NaN poses: anywhere will do, their boxes overlap nothing)
      }
    std::vector<int> ord(st.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::vector<std::pair<size_t, size_t>> todo(1, std::make_pair((size_t)0, st.size()));
    while (!todo.empty()) {
      const size_t b = todo.back().first, n = todo.back().second;
      todo.pop_back();
      if (n <= 64) continue;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (size_t i = b; i < b + n; ++i)
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], ctr[3 * ord[i] + k]); hi[k] = std::max(hi[k], ctr[3 * ord[i] + k]); }
      int ax = 0;
      if (hi[1] - lo[1] > hi[ax] - lo[ax]) ax = 1;
      if (hi[2] - lo[2] > hi[ax] - lo[ax]) ax = 2;
      std::sort(ord.begin() + b, ord.begin() + b + n, [&](int x, int y) { return ctr[3 * x + ax] < ctr[3 * y + ax] || (ctr[3 * x + ax] == ctr[3 * y + ax] && x < y); });
      const size_t nl = (n + 127) / 128 * 64;
      todo.push_back(std::make_pair(b, nl));
      todo.push_back(std::make_pair(b + nl, n - nl));
    }
    std::vector<int> sorted(st.size());
    for (size_t i = 0; i < st.size(); ++i) sorted[i] = st[ord[i]];
    st.swap(sorted);
  }
  e->n_clR = (e->n_statR + 63) / 64;
  e->n_stat_padR = std::max(e->n_clR, 1) * 64;
  e->n_cl_padR = (e->n_clR + 63) / 64 * 64 + 64;
  std::vector<float> soa((size_t)6 * e->n_stat_padR), cl((size_t)6 * e->n_cl_padR);
  for (int k = 0; k < 6; ++k) {   // pads: inverted boxes, no closed-interval test can pass
    std::fill(soa.begin() + (size_t)k * e->n_stat_padR, soa.begin() + (size_t)(k + 1) * e->n_stat_padR, k < 3 ? INFINITY : -INFINITY);
    std::fill(cl.begin() + (size_t)k * e->n_cl_padR, cl.begin() + (size_t)(k + 1) * e->n_cl_padR, k < 3 ? INFINITY : -INFINITY);
  }
  for (int i = 0; i < e->n_statR; ++i) {
    const int c = i / 64;
    for (int k = 0; k < 6; ++k) {
      const float v = bb[6 * (size_t)st[i] + k];
      soa[(size_t)k * e->n_stat_padR + i] = v;
      float &u = cl[(size_t)k * e->n_cl_padR + c];
      u = k < 3 ? std::min(u, v) : std::max(u, v);
    }
  }
  std::vector<int> stp(e->n_stat_padR, 0);
  std::copy(st.begin(), st.end(), stp.begin());
  HIPCHK(upload(e->d_stat_soaR, soa, e->stream));
  HIPCHK(upload(e->d_cl_soaR, cl, e->stream));
  HIPCHK(upload(e->d_stat_childrenR, stp, e->stream));
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
  // Validate EVERYTHING before the engine is touched: a rejected upload leaves the previous scene (host tables, device
  // buffers, have_scene / have_robot) exactly as it was.
  for (int i = 0; i < n_shapes; ++i) {
    const float *p = shape_param + 4 * i;
    if (shape_kind[i] == TOHC_BOX) {
      if (!(p[0] > 0 && p[1] > 0 && p[2] > 0)) return fail(e, TOHC_E_INVALID, "upload_scene: box %d has non-positive extents", i);
    } else if (shape_kind[i] == TOHC_SPHERE) {
      if (!(p[0] > 0)) return fail(e, TOHC_E_INVALID, "upload_scene: sphere %d has a non-positive radius", i);
    } else if (shape_kind[i] == TOHC_CYLZ) {
      if (!(p[0] > 0 && p[1] > 0)) return fail(e, TOHC_E_INVALID, "upload_scene: cylinder %d has a non-positive radius or height", i);
    } else if (shape_kind[i] == TOHC_HULL) {
      if (!shape_voff || !shape_vcnt || !vx || !vy || !vz || shape_vcnt[i] < 1 || shape_vcnt[i] > 64 || shape_voff[i] < 0 ||
          (int64_t)shape_voff[i] + shape_vcnt[i] > n_verts)
        return fail(e, TOHC_E_INVALID, "upload_scene: hull %d has a bad vertex range (1..64 vertices supported)", i);
    } else {
      return fail(e, TOHC_E_UNSUPPORTED, "upload_scene: shape %d has kind %d; boxes, spheres, z-cylinders and convex hulls are supported", i, shape_kind[i]);
    }
  }
  for (int c = 0; c < n_children; ++c)
    if (child_shape[c] < 0 || child_shape[c] >= n_shapes) return fail(e, TOHC_E_INVALID, "upload_scene: child %d references shape %d", c, child_shape[c]);
  {
    std::vector<char> owned(n_children, 0);
    for (int o = 0; o < n_objects; ++o) {
      if (obj_child_off[o] < 0 || obj_child_cnt[o] < 1 || (int64_t)obj_child_off[o] + obj_child_cnt[o] > n_children)
        return fail(e, TOHC_E_INVALID, "upload_scene: object %d has a bad child range", o);
      if (!obj_compound[o] && obj_child_cnt[o] != 1) return fail(e, TOHC_E_INVALID, "upload_scene: bare convex object %d must have one child", o);
      for (int k = 0; k < obj_child_cnt[o]; ++k) {
        if (owned[obj_child_off[o] + k]) return fail(e, TOHC_E_INVALID, "upload_scene: child %d belongs to two objects", obj_child_off[o] + k);
        owned[obj_child_off[o] + k] = 1;
      }
    }
    for (int c = 0; c < n_children; ++c)
      if (!owned[c]) return fail(e, TOHC_E_INVALID, "upload_scene: child %d belongs to no object", c);
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
  // from here on only HIP calls can fail; until the last of them has succeeded the engine has no scene
  e->have_scene = e->have_robot = e->staged = false;
  e->n_shapes = n_shapes; e->n_children = n_children; e->n_objects = n_objects; e->n_verts = n_verts;
  e->h_shape_kind.assign(shape_kind, shape_kind + n_shapes);
  e->h_shape_voff.assign(n_shapes, 0); e->h_shape_vcnt.assign(n_shapes, 0);
  e->h_shape_dims.assign((size_t)n_shapes * 3, 0.f); e->h_shape_lmin.assign((size_t)n_shapes * 3, 0.f); e->h_shape_lmax.assign((size_t)n_shapes * 3, 0.f);
  for (int i = 0; i < n_shapes; ++i) {
    const float *p = shape_param + 4 * i;
    if (shape_kind[i] == TOHC_BOX) {
      // btBoxShape ctor (setSafeMargin; dims = h - margin) followed by the checker's setMargin(0): (dims + margin) - 0
      // (btBoxShape.cpp:17-26, btBoxShape.h:87-96, bullet_collision_checker.cpp:227,244)
      float m = safeMargin(0.04f, p[0], p[1], p[2]);
      for (int k = 0; k < 3; ++k) {
        float dim = ((p[k] * 1.0f - m) + m) - 0.0f;
        e->h_shape_dims[3 * i + k] = dim; e->h_shape_lmin[3 * i + k] = -dim; e->h_shape_lmax[3 * i + k] = dim;
      }
    } else if (shape_kind[i] == TOHC_SPHERE) {
      // btSphereShape(radius): m_implicitShapeDimensions.x = radius, getMargin() = getRadius() whatever setMargin(0) did
      // (btSphereShape.h:29-67); local box = centre +- radius
      for (int k = 0; k < 3; ++k) { e->h_shape_dims[3 * i + k] = p[0]; e->h_shape_lmin[3 * i + k] = -p[0]; e->h_shape_lmax[3 * i + k] = p[0]; }
    } else if (shape_kind[i] == TOHC_CYLZ) {
      // btCylinderShapeZ(r, r, h/2): same margin round trip as the box (btCylinderShape.cpp:18-27); param = (radius, half height)
      const float he[3] = {p[0], p[0], p[1]};
      float m = safeMargin(0.04f, he[0], he[1], he[2]);
      for (int k = 0; k < 3; ++k) {
        float dim = ((he[k] * 1.0f - m) + m) - 0.0f;
        e->h_shape_dims[3 * i + k] = dim; e->h_shape_lmin[3 * i + k] = -dim; e->h_shape_lmax[3 * i + k] = dim;
      }
    } else {   // TOHC_HULL (validated above)
      int off = shape_voff[i], cnt = shape_vcnt[i];
      e->h_shape_voff[i] = off; e->h_shape_vcnt[i] = cnt;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (int k = 0; k < cnt; ++k) {
        float q[3] = {vx[off + k], vy[off + k], vz[off + k]};
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], q[a]); hi[a] = fmaxf(hi[a], q[a]); }
      }
      for (int a = 0; a < 3; ++a) { e->h_shape_lmin[3 * i + a] = lo[a]; e->h_shape_lmax[3 * i + a] = hi[a]; }
    }
  }
  // bounding sphere of every shape about the centre of its local box (k_broadphase's first pre-test stage); the radius
  // is rounded up so that the sphere certainly holds the shape
  e->h_shape_sph.assign((size_t)n_shapes * 4, 0.f);
  for (int i = 0; i < n_shapes; ++i) {
    float c[3], r2 = 0;
    for (int k = 0; k < 3; ++k) c[k] = 0.5f * (e->h_shape_lmin[3 * i + k] + e->h_shape_lmax[3 * i + k]);
    if (shape_kind[i] == TOHC_HULL) {
      for (int k = 0; k < e->h_shape_vcnt[i]; ++k) {
        const int v = e->h_shape_voff[i] + k;
        const double dx = (double)vx[v] - c[0], dy = (double)vy[v] - c[1], dz = (double)vz[v] - c[2];
        r2 = std::max(r2, (float)(dx * dx + dy * dy + dz * dz));
      }
    } else if (shape_kind[i] == TOHC_SPHERE) {
      r2 = e->h_shape_dims[3 * i] * e->h_shape_dims[3 * i];
    } else {   // box / cylinder: the corner of the local box
      for (int k = 0; k < 3; ++k) { const double h = (double)e->h_shape_lmax[3 * i + k] - c[k]; r2 += (float)(h * h); }
    }
    for (int k = 0; k < 3; ++k) e->h_shape_sph[4 * i + k] = c[k];
    e->h_shape_sph[4 * i + 3] = sqrtf(r2) * 1.00001f + 1.0e-6f;
  }
  e->h_child_shape.assign(child_shape, child_shape + n_children);
  e->h_child_obj.assign(n_children, -1); e->h_child_idx.assign(n_children, 0);
  e->h_child_localT.assign((size_t)n_children * 12, 0.f);
  for (int c = 0; c < n_children; ++c) stXf(&e->h_child_localT[12 * c], xfFrom7(child_T + 7 * c));
  e->h_obj_off.assign(obj_child_off, obj_child_off + n_objects); e->h_obj_cnt.assign(obj_child_cnt, obj_child_cnt + n_objects);
  e->children_in_object_order = true;
  for (int o = 0, run = 0; o < n_objects; ++o) { e->children_in_object_order &= obj_child_off[o] == run; run += obj_child_cnt[o]; }
  e->h_obj_compound.assign(obj_compound, obj_compound + n_objects); e->h_obj_group.assign(obj_group, obj_group + n_objects);
  e->h_obj_static_T.assign(obj_static_T, obj_static_T + (size_t)7 * n_objects);
  for (int o = 0; o < n_objects; ++o)
    for (int k = 0; k < obj_child_cnt[o]; ++k) {
      e->h_child_obj[obj_child_off[o] + k] = o;
      e->h_child_idx[obj_child_off[o] + k] = k;
    }
  e->h_allowed.assign((size_t)n_objects * n_objects, 1);
  // "poses" mode: every object moves
  e->h_obj_dynP.resize(n_objects); std::iota(e->h_obj_dynP.begin(), e->h_obj_dynP.end(), 0);
  e->h_child_dynP.resize(n_children); std::iota(e->h_child_dynP.begin(), e->h_child_dynP.end(), 0);
  e->h_dyn_childrenP = e->h_child_dynP;
  e->h_stat_childrenR = e->h_child_dynP;   // no robot yet: nothing moves in robot mode
  e->h_child_dynR.clear(); e->h_obj_dynR.clear(); e->h_dyn_childrenR.clear(); e->h_qchildrenR.clear();   // a previous robot's maps index the old scene
  e->n_dyn_objR = e->n_dyn_childrenR = 0;
  e->h_stat_childrenP.clear();
  hipStream_t s = e->stream;
  HIPCHK(upload(e->d_shape_kind, e->h_shape_kind, s)); HIPCHK(upload(e->d_shape_voff, e->h_shape_voff, s)); HIPCHK(upload(e->d_shape_vcnt, e->h_shape_vcnt, s));
  HIPCHK(upload(e->d_shape_dims, e->h_shape_dims, s)); HIPCHK(upload(e->d_shape_lmin, e->h_shape_lmin, s)); HIPCHK(upload(e->d_shape_lmax, e->h_shape_lmax, s));
  {
    // SoA vertex arrays, padded so that a 64-lane load past the last hull stays in bounds
    std::vector<float> px((size_t)n_verts + 64, 0.f), py((size_t)n_verts + 64, 0.f), pz((size_t)n_verts + 64, 0.f);
    for (int i = 0; i < n_verts; ++i) { px[i] = vx[i]; py[i] = vy[i]; pz[i] = vz[i]; }
    HIPCHK(upload(e->d_vx, px, s)); HIPCHK(upload(e->d_vy, py, s)); HIPCHK(upload(e->d_vz, pz, s));
    std::vector<F4> p4((size_t)n_verts + 64);
    for (size_t i = 0; i < p4.size(); ++i) { p4[i].x = px[i]; p4[i].y = py[i]; p4[i].z = pz[i]; p4[i].w = 0.f; }
    HIPCHK(upload(e->d_v4, p4, s));
    std::vector<float> v8;
    blockVertices(px.data(), py.data(), pz.data(), n_shapes, shape_kind, shape_voff, shape_vcnt, v8, e->h_shape_voff8);
    HIPCHK(upload(e->d_v8, v8, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(upload(e->d_child_shape, e->h_child_shape, s)); HIPCHK(upload(e->d_child_obj, e->h_child_obj, s)); HIPCHK(upload(e->d_child_idx, e->h_child_idx, s));
  HIPCHK(upload(e->d_child_localT, e->h_child_localT, s));
  HIPCHK(upload(e->d_obj_off, e->h_obj_off, s)); HIPCHK(upload(e->d_obj_cnt, e->h_obj_cnt, s)); HIPCHK(upload(e->d_obj_compound, e->h_obj_compound, s));
  HIPCHK(upload(e->d_obj_group, e->h_obj_group, s)); HIPCHK(upload(e->d_allowed, e->h_allowed, s));
  HIPCHK(upload(e->d_obj_dynP, e->h_obj_dynP, s)); HIPCHK(upload(e->d_child_dynP, e->h_child_dynP, s)); HIPCHK(upload(e->d_dyn_childrenP, e->h_dyn_childrenP, s));
  HIPCHK(upload(e->d_stat_childrenP, std::vector<int>(1, 0), s)); HIPCHK(upload(e->d_stat_soaP, std::vector<float>(6 * 64, 0.f), s));
  HIPCHK(hipStreamSynchronize(s));
  const int rc = refreshStatic(e);
  e->have_scene = rc == TOHC_OK;
  return rc;
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
  // validate first: a rejected upload leaves the previous robot in place, tables and flags untouched
  const int newK = n_joint_inds + ((affine & 1) != 0) + ((affine & 2) != 0) + ((affine & 4) != 0) + ((affine & 8) != 0);
  if (newK <= 0) return fail(e, TOHC_E_INVALID, "upload_robot: no active degrees of freedom");
  {
    std::vector<char> taken(e->n_objects, 0);
    for (int i = 0; i < n_links; ++i) {
      if (parent[i] >= i || parent[i] < -1) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: parents must precede children", i);
      if (dof[i] >= n_dof) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: dof %d out of range", i, dof[i]);
      if (dof[i] == TOHC_DOF_GRABBED && (jtype[i] != TOHC_J_FIXED || parent[i] < 0))
        return fail(e, TOHC_E_INVALID, "upload_robot: link %d: a grabbed body's link is a fixed link with a grabber (parent)", i);
      if (jtype[i] != TOHC_J_FIXED && jtype[i] != TOHC_J_REVOLUTE && jtype[i] != TOHC_J_PRISMATIC)
        return fail(e, TOHC_E_INVALID, "upload_robot: link %d: unknown joint type %d", i, jtype[i]);
      if (link_obj[i] >= e->n_objects) return fail(e, TOHC_E_INVALID, "upload_robot: link %d: object %d out of range", i, link_obj[i]);
      if (link_obj[i] >= 0) {
        if (taken[link_obj[i]]) return fail(e, TOHC_E_INVALID, "upload_robot: object %d is attached to two links", link_obj[i]);
        taken[link_obj[i]] = 1;
      }
    }
    for (int a = 0; a < n_joint_inds; ++a)
      if (joint_inds[a] < 0 || joint_inds[a] >= n_dof) return fail(e, TOHC_E_INVALID, "upload_robot: joint_inds[%d] = %d out of range", a, joint_inds[a]);
  }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
  e->have_robot = e->staged = false;   // until the last HIP call below has succeeded
  e->n_links = n_links; e->n_joint_inds = n_joint_inds; e->affine = affine; e->n_dof = n_dof;
  e->K = newK;
  e->h_parent.assign(parent, parent + n_links); e->h_jtype.assign(jtype, jtype + n_links); e->h_dof.assign(dof, dof + n_links);
  e->h_link_obj.assign(link_obj, link_obj + n_links);
  e->h_dof_col.assign(n_links, -1);
  for (int i = 0; i < n_links; ++i) {
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
    e->h_obj_dynR[o] = nd; e->h_link_dyn[i] = nd; ++nd;
    bool affected = affine > 0 || dof[i] == TOHC_DOF_GRABBED;   // GetAffectedLinks lists grabbed links unconditionally (:120-131)
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
  e->h_stat_childrenR.clear();
  for (int c = 0; c < e->n_children; ++c)
    if (e->h_child_dynR[c] < 0) e->h_stat_childrenR.push_back(c);
  // ancestors of each link, root first (k_fk composes every chain in that order)
  e->h_chain_off.assign(n_links + 1, 0); e->h_chain.clear();
  e->shared_columns = false;
  for (int i = 0; i < n_links; ++i) {
    std::vector<int> up;
    for (int j = i; j >= 0; j = parent[j]) up.push_back(j);
    for (auto it = up.rbegin(); it != up.rend(); ++it) e->h_chain.push_back(*it);
    e->h_chain_off[i + 1] = (int)e->h_chain.size();
    // two moving joints of one chain on the same optimisation column (mimic joints): their Jacobian columns add
    // BEFORE the product with the normal, which the lane-per-joint epilogue of k_finish does not do
    for (size_t a = (size_t)e->h_chain_off[i]; a < e->h_chain.size(); ++a)
      for (size_t b = a + 1; b < e->h_chain.size(); ++b) {
        const int ja = e->h_chain[a], jb = e->h_chain[b];
        if (e->h_jtype[ja] != TOHC_J_FIXED && e->h_jtype[jb] != TOHC_J_FIXED && e->h_dof_col[ja] >= 0 && e->h_dof_col[ja] == e->h_dof_col[jb]) e->shared_columns = true;
      }
  }
  hipStream_t s = e->stream;
  {
    std::vector<JacJoint> jj;
    std::vector<int> j0(n_links), j1(n_links);
    for (int i = 0; i < n_links; ++i) {
      j0[i] = (int)jj.size();
      for (int c = e->h_chain_off[i]; c < e->h_chain_off[i + 1]; ++c) {
        const int j = e->h_chain[c], k = e->h_dof_col[j];
        if (jtype[j] == TOHC_J_FIXED || k < 0 || k >= n_joint_inds) continue;
        JacJoint r;
        memset(&r, 0, sizeof r);
        r.link = j; r.col = k; r.revolute = jtype[j] == TOHC_J_REVOLUTE ? 1 : 0; r.mult = mult[j];
        jj.push_back(r);
      }
      j1[i] = (int)jj.size();
    }
    std::vector<int4> cj(e->n_children);
    for (int c = 0; c < e->n_children; ++c) {
      const int l = e->h_obj_jaclink[e->h_child_obj[c]];
      cj[c] = make_int4(l, l >= 0 ? j0[l] : 0, l >= 0 ? j1[l] : 0, 0);
    }
    if (jj.empty()) { JacJoint r; memset(&r, 0, sizeof r); jj.push_back(r); }
    HIPCHK(upload(e->d_jac_joint, jj, s)); HIPCHK(upload(e->d_child_jac, cj, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(upload(e->d_parent, e->h_parent, s)); HIPCHK(upload(e->d_jtype, e->h_jtype, s)); HIPCHK(upload(e->d_dof, e->h_dof, s));
  HIPCHK(upload(e->d_dof_col, e->h_dof_col, s)); HIPCHK(upload(e->d_link_dyn, e->h_link_dyn, s)); HIPCHK(upload(e->d_obj_jaclink, e->h_obj_jaclink, s));
  HIPCHK(upload(e->d_link_obj, e->h_link_obj, s)); HIPCHK(upload(e->d_chain_off, e->h_chain_off, s)); HIPCHK(upload(e->d_chain, e->h_chain, s));
  HIPCHK(upload(e->d_T_origin, T_origin, (size_t)7 * n_links, s)); HIPCHK(upload(e->d_axis, axis, (size_t)3 * n_links, s));
  HIPCHK(upload(e->d_mult, mult, (size_t)n_links, s)); HIPCHK(upload(e->d_offs, offs, (size_t)n_links, s));
  {
    std::vector<double> q(std::max(n_dof, 1), 0.0);
    for (int i = 0; i < n_dof; ++i) q[i] = q_robot[i];
    HIPCHK(upload(e->d_q_robot, q, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(upload(e->d_obj_dynR, e->h_obj_dynR, s)); HIPCHK(upload(e->d_child_dynR, e->h_child_dynR, s));
  HIPCHK(upload(e->d_dyn_childrenR, e->h_dyn_childrenR, s)); e->h_qinfoR = queryInfo(e, e->h_qchildrenR, e->h_child_dynR);
  HIPCHK(upload(e->d_qchildrenR, e->h_qinfoR, s));
  HIPCHK(hipStreamSynchronize(s));
  const int rc = refreshStatic(e);
  e->have_robot = rc == TOHC_OK;
  return rc;
}

int tohc_set_allowed_matrix(tohc_engine *e, int n_objects, const uint8_t *allowed) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "set_allowed_matrix: upload the scene first");
  if (n_objects != e->n_objects || !allowed) return fail(e, TOHC_E_INVALID, "set_allowed_matrix: expected a %d x %d matrix", e->n_objects, e->n_objects);
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
  e->h_allowed.assign(allowed, allowed + (size_t)n_objects * n_objects);
  HIPCHK(upload(e->d_allowed, e->h_allowed, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return TOHC_OK;
}

int tohc_set_contact_distance(tohc_engine *e, float d) {
  if (!e) return TOHC_E_INVALID;
  if (!(d >= 0)) return fail(e, TOHC_E_INVALID, "set_contact_distance: negative or NaN distance");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
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
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));   // the queues are re-allocated at the next enqueue
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
  s.vx = e->d_vx.p; s.vy = e->d_vy.p; s.vz = e->d_vz.p; s.v4 = e->d_v4.p; s.v8 = e->d_v8.p;
  s.child_shape = e->d_child_shape.p; s.child_obj = e->d_child_obj.p; s.child_idx = e->d_child_idx.p; s.child_localT = e->d_child_localT.p;
  s.obj_off = e->d_obj_off.p; s.obj_cnt = e->d_obj_cnt.p; s.obj_compound = e->d_obj_compound.p; s.obj_group = e->d_obj_group.p;
  s.allowed = e->d_allowed.p; s.stat_xf = e->d_stat_xf.p; s.stat_aabb = e->d_stat_aabb.p; s.child_desc = e->d_child_desc.p; s.stat_sph = e->d_stat_sph.p;
  return s;
}
static DevRobot devRobot(tohc_engine *e) {
  DevRobot r;
  memset(&r, 0, sizeof r);
  r.n_links = e->n_links; r.K = e->K; r.n_joint_inds = e->n_joint_inds; r.affine = e->affine;
  r.parent = e->d_parent.p; r.jtype = e->d_jtype.p; r.dof = e->d_dof.p; r.dof_col = e->d_dof_col.p; r.link_dyn = e->d_link_dyn.p;
  r.T_origin = e->d_T_origin.p; r.axis = e->d_axis.p; r.mult = e->d_mult.p; r.offs = e->d_offs.p; r.q_robot = e->d_q_robot.p;
  r.obj_jaclink = e->d_obj_jaclink.p; r.link_obj = e->d_link_obj.p; r.chain_off = e->d_chain_off.p; r.chain = e->d_chain.p;
  r.child_jac = e->d_child_jac.p; r.jac_joint = e->d_jac_joint.p;
  r.fast_jac = (!e->shared_columns && e->K <= TOHC_FAST_JAC_MAXK) ? 1 : 0;
  memcpy(r.base_T, e->base_T, sizeof r.base_T);
  memcpy(r.rot_axis, e->rot_axis, sizeof r.rot_axis);
  return r;
}

static DynMap dynMapRobot(tohc_engine *e) {
  DynMap dm;
  dm.obj_dyn = e->d_obj_dynR.p; dm.child_dyn = e->d_child_dynR.p; dm.n_dyn_obj = e->n_dyn_objR; dm.n_dyn_children = e->n_dyn_childrenR;
  dm.dyn_children = e->d_dyn_childrenR.p; dm.stat_children = e->d_stat_childrenR.p; dm.stat_soa = e->d_stat_soaR.p;
  dm.n_stat = e->n_statR; dm.n_stat_pad = e->n_stat_padR;
  dm.cl_soa = e->d_cl_soaR.p; dm.n_cl = e->n_clR; dm.n_cl_pad = e->n_cl_padR;
  return dm;
}
static DynMap dynMapPoses(tohc_engine *e) {
  DynMap dm;
  dm.obj_dyn = e->d_obj_dynP.p; dm.child_dyn = e->d_child_dynP.p; dm.n_dyn_obj = e->n_objects; dm.n_dyn_children = e->n_children;
  dm.dyn_children = e->d_dyn_childrenP.p; dm.stat_children = e->d_stat_childrenP.p; dm.stat_soa = e->d_stat_soaP.p;
  dm.n_stat = 0; dm.n_stat_pad = 64;
  dm.cl_soa = e->d_stat_soaP.p; dm.n_cl = 0; dm.n_cl_pad = 64;
  return dm;
}

// -----------------------------------------------------------------------------------------------------------------
// stage -> enqueue -> sync -> fetch.  The one-shot batch calls are exactly these four in a row.
// -----------------------------------------------------------------------------------------------------------------
static int syncEngineImpl(tohc_engine *e) {
  hipStream_t s = e->stream;
  {
    const int rc = launchTail(e);   // passes whose tail is still to be launched
    if (rc != TOHC_OK) return rc;
  }
  for (auto &ln : e->lane) HIPCHK(hipStreamSynchronize(ln.stream));
  HIPCHK(hipStreamSynchronize(e->tail_stream));
  if (e->pending.empty()) { HIPCHK(hipStreamSynchronize(s)); return TOHC_OK; }
  // ONE copy: the record counter (element 0) and the slots in use, which are 0 .. pending - 1 (a Counters is 2 KB)
  HIPCHK(hipMemcpyAsync(e->h_cnt.p, e->d_cnt.p, (1 + e->pending.size()) * sizeof(Counters), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  bool overflow = false;
  unsigned int worst_pairs = 0;
  tohc_counters &c = e->counters;
  c.aabb_tests = c.pair_evals = c.epa_evals = c.gjk_evals = 0;
  c.gjk_iters = c.pen_supports = c.epa_supports = c.epa_redone = c.pen_refiled = 0;
  c.ms_last_call = 0;
  for (const auto &pd : e->pending) {
    const Counters &k = e->h_cnt.p[1 + pd.slot];
    const unsigned int segCap = (unsigned int)(e->max_pairs / TOHC_NSHARD);
    unsigned int np = 0, ne = 0, ng = 0;
    int64_t gi = 0, ps = 0, es = 0, er = 0, pr = 0;
    overflow |= k.overflow != 0;
    for (int sh = 0; sh < TOHC_NSHARD; ++sh) {
      const CounterShard &cs = k.sh[sh];
      // (n_pen may pass its shard's capacity without loss: k_tail_pen re-files such a pair under its block's shard and
      // raises Counters::overflow itself when that one is full too)
      overflow |= cs.n_pairs > segCap || cs.n_exact > segCap || cs.n_epa > segCap || cs.n_hits > segCap;
      np += cs.n_cand; ng += cs.n_pairs; ne += cs.n_epa;
      gi += cs.n_gjk_steps; ps += cs.n_pen_sups; es += cs.n_epa_sups; er += cs.n_epa_redone; pr += cs.n_pen_refiled;
    }
    c.gjk_iters += gi; c.pen_supports += ps; c.epa_supports += es;
    c.cum_gjk_iters += gi; c.cum_pen_supports += ps; c.cum_epa_supports += es;
    c.epa_redone += er; c.cum_epa_redone += er;
    c.pen_refiled += pr; c.cum_pen_refiled += pr;
    worst_pairs = std::max(worst_pairs, ng);
    if (pd.kind >= 0) { e->gjk_hint[pd.kind].units = pd.units; e->gjk_hint[pd.kind].pairs = ng; }
    c.aabb_tests += pd.aabb_tests; c.pair_evals += np; c.epa_evals += ne; c.gjk_evals += ng;
    c.cum_aabb_tests += pd.aabb_tests; c.cum_pair_evals += np; c.cum_epa_evals += ne; c.cum_gjk_evals += ng;
    float ms[NEV - 1] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t *ev = &e->events[(size_t)pd.slot * NEV];
    for (int i = 0; i < NEV - 1 && e->timing; ++i)
      if (i != 4 && i != 6) HIPCHK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));   // 4 -> 5 and 6 -> 7 are waits between streams
    c.cum_ms_fk += ms[0]; c.cum_ms_children += ms[1]; c.cum_ms_broadphase += ms[2]; c.cum_ms_gjk += ms[3]; c.cum_ms_epa += ms[5];
    c.cum_ms_finish += ms[7];
    // enqueues overlap on the lanes: the device time of the call is the span from the first start to the last end
    float span = 0;
    HIPCHK(hipEventElapsedTime(&span, e->events[(size_t)e->pending.front().slot * NEV], ev[NEV - 1]));
    c.ms_last_call = std::max(c.ms_last_call, (double)span);
    c.cum_launches += 1;
  }
#if defined(TOHC_EPA_TIMELINE)
  if (e->d_dbg.p) {
    unsigned long long h[256];
    HIPCHK(hipMemcpy(h, e->d_dbg.p, sizeof h, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->d_dbg.p, 0, sizeof h));
    fprintf(stderr, "[epa timeline] wavefront lifetimes, 25 us bins:");
    for (int k = 0; k < 32; ++k) fprintf(stderr, " %llu", h[160 + k]);
    fprintf(stderr, "\n[epa timeline] round durations, 10 us bins:");
    for (int k = 0; k < 32; ++k) fprintf(stderr, " %llu", h[192 + k]);
    fprintf(stderr, "\n[epa timeline] us spent in rounds by when they began in their wavefront's life, 25 us bins:");
    for (int k = 0; k < 32; ++k) fprintf(stderr, " %llu", h[224 + k] / 100ull);
    fprintf(stderr, "\n");
  }
#endif
#if defined(TOHC_EPA_PROF)
  if (e->d_dbg.p) {
    unsigned long long h[160];
    HIPCHK(hipMemcpy(h, e->d_dbg.p, sizeof h, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->d_dbg.p, 0, sizeof h));
    if (h[0]) {
      fprintf(stderr, "[epa prof] evals %llu  mean wall %.2f us  max wall %.2f us  mean cycles %.0f\n", h[0], h[1] / (double)h[0] / 100.0, h[3] / 100.0, h[2] / (double)h[0]);
      const char *nm[8] = {"init", "support", "expand(x3)", "findbest", "enclose", "newface x4", "evaluate", "makeHorizon"};
      fprintf(stderr, "   expansions through the exact walk %llu, ended by a face reached twice %llu\n", h[4], h[5]);
      fprintf(stderr, "   support vertices added (histogram 0..63):");
      for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", h[32 + k]);
      fprintf(stderr, "\n   highest face slot used / 2 (histogram 0..63):");
      for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", h[96 + k]);
      fprintf(stderr, "\n");
      for (int k = 0; k < 8; ++k) fprintf(stderr, "   %-12s calls/eval %.2f  cycles/eval %.0f  cycles/call %.0f\n", nm[k], h[16 + k] / (double)h[0], h[8 + k] / (double)h[0], h[16 + k] ? h[8 + k] / (double)h[16 + k] : 0.0);
    }
  }
#endif
  e->pending.clear();
  const unsigned int nrec = *e->h_nrec.p;
  c.contacts = nrec;
  if (overflow || nrec > e->max_records)
    return fail(e, TOHC_E_CAPACITY, "capacity exceeded: %u pairs in one enqueue (max %zu, split over 16 shards), %u records (max %zu); raise with tohc_set_capacity",
                worst_pairs, e->max_pairs, nrec, e->max_records);
  return TOHC_OK;
}

static int syncEngine(tohc_engine *e) {
  const bool had = !e->pending.empty();
  const int rc = syncEngineImpl(e);
  if (rc != TOHC_OK) { e->slab_state = -1; e->pending.clear(); }   // overflow or a HIP error: nothing of this slab may be fetched
  else if (had) e->slab_state = 1;
  return rc;
}

static int64_t countAabbTests(const tohc_engine *e, const std::vector<int> &qchildren, const CallParams &cp) {
  // tuples the broadphase examines = units x (query child, other child) combinations that pass the object filters
  int64_t per = 0;
  for (int q : qchildren) {
    int Q = e->h_child_obj[q];
    for (int X = 0; X < e->n_objects; ++X) {
      if (X == Q && !cp.cast) continue;   // (the swept temporary object is not in the world: no self to skip)
      int g = e->h_obj_group[X];
      if (cp.cast ? !(g & TOHC_GROUP_KINBODY) : !(g & (short)cp.filter_mask)) continue;
      if (!e->h_allowed[(size_t)Q * e->n_objects + X]) continue;
      per += e->h_obj_cnt[X];
    }
  }
  return per * cp.n_cast_sets;
}

// broadphase -> GJK -> EPA for one enqueue; `ev` = this enqueue's events [2..5] are recorded here
#ifndef TOHC_GJK_GRID
#define TOHC_GJK_GRID 16384   // lane-per-pair blocks of one wavefront: many small blocks, so the hardware dispatcher balances the uneven runs (4096: 0.51 ms per launch, 16384: 0.37)
#endif
#ifndef TOHC_GJK16_MAX_PAIRS
#define TOHC_GJK16_MAX_PAIRS 16384   // k_gjk<16> instead of k_gjk<1> up to this many pairs in the pass (tohc_engine::gjk_adapt)
#endif
#ifndef TOHC_DIST_BLOCKS
#define TOHC_DIST_BLOCKS 256   // blocks of k_tail_epa that run the Distance fall-back
#endif
#ifndef TOHC_EPA_GRID
#define TOHC_EPA_GRID 2048
#endif
#ifndef TOHC_EPA_G
#define TOHC_EPA_G 8   // lanes per pair in k_tail_epa (8 pairs per wavefront on the small pools: measured 0.45 ms per launch against 0.67 for 16 lanes on the full pools)
#endif
#ifndef TOHC_FINISH_G
#define TOHC_FINISH_G 4   // lanes per hit in k_finish: most of a hit's work is the same for every lane of its group (measured per launch: 16 lanes 0.38 ms, 8: 0.29, 4: 0.22, 2: 0.23, 1: 0.20 but 73 KB of LDS rows per block at K = 18)
#endif
#ifndef TOHC_FINISH_BLOCKS_PER_CU_DISCRETE
#define TOHC_FINISH_BLOCKS_PER_CU_DISCRETE 4   // the discrete pass's kernel: ~108 VGPRs, four waves per SIMD
#endif
#ifndef TOHC_FINISH_BLOCKS_PER_CU
#define TOHC_FINISH_BLOCKS_PER_CU 2   // with TOHC_FINISH_WAVES 2: two waves per SIMD = 2 blocks of 256 per CU, the whole grid resident at once (measured step: 1 block 2.41 ms, 2: 2.31, 3: 2.35, 4: 2.34, 8: 2.32)
#endif
// The tail of every open pass: ONE k_tail_pen and ONE k_tail_epa launch over all their penetration queues, their
// record ranges reserved in enqueue order (k_tail_range), then each pass's k_finish on its own lane, side by side.
static int launchTail(tohc_engine *e) {
  if (e->open.empty()) return TOHC_OK;
  std::vector<tohc_engine::OpenHead> heads;
  heads.swap(e->open);   // (whatever fails below, the passes are not launched twice)
  // a tail of one pass runs on that pass's own lane (tails of different lanes then overlap, as do their heads); a
  // shared tail on the tail stream, after every head it serves
  hipStream_t ts = heads.size() == 1 ? e->lane[heads[0].lane].stream : e->tail_stream;
  TailArgs ta;
  memset(&ta, 0, sizeof ta);
  ta.n_heads = (int)heads.size();
  bool work = false;
  for (size_t i = 0; i < heads.size(); ++i) {
    ta.h[i] = heads[i].a;
    work |= !heads[i].empty;
    HIPCHK(hipStreamWaitEvent(ts, e->lane[heads[i].lane].head_done, 0));
  }
  // the tail's time is booked on the last pass; the others get a zero-length pair of events
  for (size_t i = 0; i + 1 < heads.size(); ++i) {
    hipEvent_t *ev = &e->events[(size_t)heads[i].slot * NEV];
    HIPCHK(recEv(e, ev, 5, ts));
    HIPCHK(recEv(e, ev, 6, ts));
  }
  hipEvent_t *evl = &e->events[(size_t)heads.back().slot * NEV];
  HIPCHK(recEv(e, evl, 5, ts));
  if (work) {
    const unsigned int nh = (unsigned int)ta.n_heads;
    const unsigned int gridN = 256 * 8;
    if (nh == 1) {
      const bool cast = ta.h[0].cp.cast != 0;
      if (e->pen_group == 4) hipLaunchKernelGGL((k_tail_pen<4, 1, 2>), dim3(gridN), dim3(64), 0, ts, ta);
      else if (cast) hipLaunchKernelGGL((k_tail_pen<1, 1, 1>), dim3(gridN), dim3(64), 0, ts, ta);
      else hipLaunchKernelGGL((k_tail_pen<1, 1, 0>), dim3(gridN), dim3(64), 0, ts, ta);
      // (+ the blocks of the Distance fall-back at the head of the grid, one lane per pair)
      ta.n_dist_blocks = e->pen_group == 1 ? (e->epa_block == 16 ? 4 * TOHC_DIST_BLOCKS : TOHC_DIST_BLOCKS) : 0;
      const unsigned int nd = ta.n_dist_blocks;
      if (e->group == 64) hipLaunchKernelGGL((k_tail_epa<64, 64, 1>), dim3(gridN + nd), dim3(64), 0, ts, ta);
      else if (e->epa_block == 16) hipLaunchKernelGGL((k_tail_epa<16, 16, 1>), dim3(gridN * 4 + nd), dim3(16), 0, ts, ta);  // one group per wavefront
      else if (cast) hipLaunchKernelGGL((k_tail_epa<TOHC_EPA_G, 64, 1, 1>), dim3(TOHC_EPA_GRID + nd), dim3(64), 0, ts, ta);
      else hipLaunchKernelGGL((k_tail_epa<TOHC_EPA_G, 64, 1, 0>), dim3(TOHC_EPA_GRID + nd), dim3(64), 0, ts, ta);
    } else {   // nh == TOHC_MAX_HEADS == 2
      if (e->pen_group == 4) hipLaunchKernelGGL((k_tail_pen<4, 2, 2>), dim3(gridN * 2), dim3(64), 0, ts, ta);
      else hipLaunchKernelGGL((k_tail_pen<1, 2, 2>), dim3(gridN * 2), dim3(64), 0, ts, ta);
      ta.n_dist_blocks = e->pen_group == 1 ? (e->epa_block == 16 ? 4 * TOHC_DIST_BLOCKS : TOHC_DIST_BLOCKS) : 0;
      const unsigned int nd = ta.n_dist_blocks;
      if (e->group == 64) hipLaunchKernelGGL((k_tail_epa<64, 64, 2>), dim3(gridN * 2 + nd), dim3(64), 0, ts, ta);
      else if (e->epa_block == 16) hipLaunchKernelGGL((k_tail_epa<16, 16, 2>), dim3(gridN * 4 * 2 + nd), dim3(16), 0, ts, ta);
      else hipLaunchKernelGGL((k_tail_epa<TOHC_EPA_G, 64, 2>), dim3(TOHC_EPA_GRID * 2 + nd), dim3(64), 0, ts, ta);
    }
  }
  // A hinge assembly may still be reading the previous slab (tohc_fetch_hinges_begin): the broadphase, GJK and the
  // penetration kernels of this one run beside it, only what writes records and gradient rows (k_finish, behind the
  // range event) waits for it
  if (e->slab_read_pending) HIPCHK(hipStreamWaitEvent(ts, e->ev_slab_read, 0));
  if (e->have_ranged) HIPCHK(hipStreamWaitEvent(ts, e->tail_ranged, 0));   // record ranges are reserved in enqueue order
  hipLaunchKernelGGL(k_tail_range, dim3(1), dim3(1), 0, ts, ta);   // also for passes without work: they reserve an empty range
  HIPCHK(recEv(e, evl, 6, ts));
  HIPCHK(hipEventRecord(e->tail_ranged, ts));
  e->have_ranged = true;
  for (const auto &h : heads) {
    tohc_engine::Lane &ln = e->lane[h.lane];
    hipEvent_t *ev = &e->events[(size_t)h.slot * NEV];
    HIPCHK(hipStreamWaitEvent(ln.stream, e->tail_ranged, 0));
    HIPCHK(recEv(e, ev, 7, ln.stream));
    if (!h.empty) {
      // LDS: one pair of gradient rows per group -- only the lane-per-joint epilogue (fast_jac: K <= TOHC_FAST_JAC_MAXK, so at
      // most 64 KB for groups of 4) touches them; the lane-per-column fall-back (wide K, mimic joints) asks for none
      const size_t Kr = h.a.rb.fast_jac ? (size_t)std::max(h.a.rb.K, 1) : 0;
      const bool castPass = h.a.cp.cast != 0;
      if (e->group == 64) {
        if (castPass) hipLaunchKernelGGL((k_finish<64, true>), dim3(256 * 2), dim3(256), (size_t)(256 / 64) * 2 * Kr * sizeof(double), ln.stream, h.a);
        else hipLaunchKernelGGL((k_finish<64, false>), dim3(256 * 2), dim3(256), (size_t)(256 / 64) * 2 * Kr * sizeof(double), ln.stream, h.a);
      } else if (castPass) {
        hipLaunchKernelGGL((k_finish<TOHC_FINISH_G, true>), dim3(256 * TOHC_FINISH_BLOCKS_PER_CU), dim3(256), (size_t)(256 / TOHC_FINISH_G) * 2 * Kr * sizeof(double), ln.stream, h.a);
      } else {
        hipLaunchKernelGGL((k_finish<TOHC_FINISH_G, false>), dim3(256 * TOHC_FINISH_BLOCKS_PER_CU_DISCRETE), dim3(256), (size_t)(256 / TOHC_FINISH_G) * 2 * Kr * sizeof(double), ln.stream, h.a);
      }
    }
    HIPCHK(recEv(e, ev, 8, ln.stream));
    HIPCHK(hipEventRecord(ln.done, ln.stream));
  }
  HIPCHK(hipGetLastError());
  return TOHC_OK;
}

// One pass's head in three steps, so that two passes can share ONE broadphase launch (enqueueBoth): narrowPrepare (the
// lane, its queues, the counters, the kernel arguments), narrowBroadphase, narrowFinish (k_exact, k_gjk, k_gjk_cont, the
// books, the tail).
struct NarrowLaunch {
  int slot = 0, laneIdx = 0, nQ = 0;
  bool wantJ = false, noSlab = false, append = false, robotMode = false, unitAll = false, gjk16 = false;
  CallParams cp;
  hipStream_t s = nullptr;
  hipEvent_t *ev = nullptr;
  tohc_engine::OpenHead oh;
  CullArgs ca;
  size_t blocks = 0;
  const std::vector<int> *h_qchildren = nullptr;
};
static int narrowPrepare(tohc_engine *e, const DynMap &dm, const int *d_qchildren, const std::vector<int> &h_qchildren, CallParams cp, bool robotMode,
                         int slot, bool append, NarrowLaunch &L) {
  const int nQ = (int)h_qchildren.size();
  const int K = robotMode ? e->K : 1;
  // one slab, one layout: the enqueue that starts a slab fixes whether it carries gradient rows (and of which robot
  // mode); an appended pass must agree, or fetch would hand out rows that were never written / drop rows that were
  const bool wantJ = robotMode && cp.want_jac;
  const bool noSlab = e->slab_state == 0 && e->pending.empty();
  if (append && !noSlab) {
    if (e->slab_state < 0) return fail(e, TOHC_E_STATE, "append: the slab this would add to overflowed (or its sync failed); start a new one");
    if (e->last_robot_mode != robotMode || e->last_want_jac != wantJ)
      return fail(e, TOHC_E_INVALID, "append: want_jacobians / query mode differ from the enqueue that started the slab");
  }
  if (!append) {   // passes of the previous slab go out before this one resets the record counter
    const int rc = launchTail(e);
    if (rc != TOHC_OK) return rc;
  }
  const int laneIdx = slot % e->n_lanes;
  for (const auto &h : e->open)
    if (h.lane == laneIdx) {   // the lane's queues still belong to an open pass: its tail goes out first
      const int rc = launchTail(e);
      if (rc != TOHC_OK) return rc;
      break;
    }
  tohc_engine::Lane &ln = e->lane[laneIdx];
  hipStream_t s = ln.stream;
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  // a new slab starts only after everything that wrote the old one
  if (!append && slot > 0)
    for (auto &other : e->lane)
      if (&other != &ln) HIPCHK(hipStreamWaitEvent(s, other.done, 0));
  // the pass's counters, and the record counter when a slab starts (element 0 of d_cnt: with slot 0 one fill clears both).
  // Before the wait for FK: nothing FK writes is touched, so the fills run beside it instead of after it.
  if (!append && slot == 0) HIPCHK(hipMemsetAsync(e->d_cnt.p, 0, 2 * sizeof(Counters), s));
  else {
    HIPCHK(hipMemsetAsync(e->d_cnt.p + 1 + slot, 0, sizeof(Counters), s));
    if (!append) HIPCHK(hipMemsetAsync(e->d_nrec.p, 0, sizeof(unsigned int), s));
  }
  HIPCHK(hipStreamWaitEvent(s, e->ev_fk, 0));   // pose tables of this call (FK / k_dyn_children on the main stream)
  cp.seg_cap = (unsigned int)(e->max_pairs / TOHC_NSHARD); cp.max_records = (unsigned int)e->max_records;
  cp.dbg_max_iter = getenv("TOHC_DBG_MAXITER") ? atoi(getenv("TOHC_DBG_MAXITER")) : 1000;
  cp.epa_small_v = e->epa_small_v; cp.epa_small_f = e->epa_small_f;
  if (cp.t_disc <= 0) cp.t_disc = cp.T;   // (no window: the discrete pass covers every staged timestep)
  const size_t qcap = (size_t)cp.seg_cap * TOHC_NSHARD;
  HIPCHK(ln.pairs.reserve(qcap)); HIPCHK(ln.epaq.reserve(qcap)); HIPCHK(ln.hits.reserve(qcap));
  cp.pen_cap = std::min(cp.seg_cap / TOHC_PEN_DIV, e->pen_cap_limit);
  HIPCHK(ln.penq.reserve((size_t)(cp.seg_cap / TOHC_PEN_DIV) * TOHC_NSHARD));
  HIPCHK(ln.distq.reserve(qcap));
  {
    const tohc_engine::GjkHint &hint = e->gjk_hint[cp.cast ? 1 : 0];
    L.gjk16 = e->gjk_group == 16 || (e->gjk_group == 1 && e->gjk_adapt && hint.units == cp.n_cast_sets && hint.pairs <= TOHC_GJK16_MAX_PAIRS);
  }
  cp.gjk_split = (e->gjk_group == 1 && !L.gjk16) ? e->gjk_split : 0;
  if (cp.gjk_split > 0) HIPCHK(ln.contq.reserve((size_t)(cp.seg_cap / TOHC_CONT_DIV) * TOHC_NSHARD));
  if (e->defer_exact) HIPCHK(ln.cands.reserve(qcap));
#if defined(TOHC_EPA_PROF) || defined(TOHC_EPA_TIMELINE)
  if (!e->d_dbg.p) { HIPCHK(e->d_dbg.reserve(256)); HIPCHK(hipMemset(e->d_dbg.p, 0, 256 * sizeof(unsigned long long))); }
#endif
  HIPCHK(e->d_records.reserve(e->max_records));
  HIPCHK(e->d_grads.reserve(e->max_records * 2 * (size_t)K));
  HIPCHK(recEv(e, ev, 2, s));
  L.slot = slot; L.laneIdx = laneIdx; L.nQ = nQ; L.wantJ = wantJ; L.noSlab = noSlab; L.append = append; L.robotMode = robotMode;
  L.s = s; L.ev = ev; L.h_qchildren = &h_qchildren;
  tohc_engine::OpenHead &oh = L.oh;
  oh.slot = slot; oh.lane = laneIdx; oh.empty = !(nQ > 0 && cp.n_cast_sets > 0);
  const bool sep1 = cp.gap == 0 && cp.cast;  // poses mode keeps the end-of-sweep poses in their own tables
  NarrowArgs &a = oh.a;
  memset(&a, 0, sizeof a);
  a.sc = devScene(e); a.dm = dm; a.rb = devRobot(e); a.cp = cp;
  a.dynXf0 = e->d_dynXf.p; a.dynXf1 = sep1 ? e->d_dynXf1.p : e->d_dynXf.p;
  a.linkQT = robotMode ? e->d_linkQT.p : nullptr;
  a.linkZ = robotMode ? e->d_linkZ.p : nullptr;
  a.dofs = robotMode ? e->d_dofs.p : nullptr;
  a.pairs = ln.pairs.p; a.cands = ln.cands.p; a.epaq = ln.epaq.p; a.penq = ln.penq.p; a.distq = ln.distq.p; a.contq = ln.contq.p; a.hits = ln.hits.p; a.cnt = e->d_cnt.p + 1 + slot; a.n_records = e->d_nrec.p; a.rec_base = e->d_rec_base.p + slot;
  a.records = e->d_records.p; a.grads = e->d_grads.p; a.dbg = e->d_dbg.p;
  L.cp = cp;
  memset(&L.ca, 0, sizeof L.ca);
  L.unitAll = false;
  L.blocks = 0;
  if (!oh.empty) {
    L.blocks = (size_t)cp.n_cast_sets * nQ;
    if (L.blocks > 0x7fffffffull) return fail(e, TOHC_E_CAPACITY, "broadphase grid too large (%zu blocks)", L.blocks);
    CullArgs &ca = L.ca;
    ca.a = a; ca.qchildren = d_qchildren; ca.nQ = nQ;
    ca.dynAabb0 = e->d_dynAabb.p; ca.dynAabb1 = sep1 ? e->d_dynAabb1.p : e->d_dynAabb.p;
    if (cp.cast && dm.n_dyn_children > 0) {   // the swept children's second boxes as the reference rounds them (k_cast_boxes)
      const size_t tot = (size_t)cp.n_cast_sets * dm.n_dyn_children;
      HIPCHK(ln.castAabb.reserve((size_t)std::max<int64_t>(e->last_sets, 1) * dm.n_dyn_children * 6));
      hipLaunchKernelGGL(k_cast_boxes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a.sc, dm, cp, a.dynXf0, a.dynXf1, ln.castAabb.p);
      ca.dynAabb1 = ln.castAabb.p;
    }
    ca.dynSph0 = e->d_dynSph.p;
    ca.pretest = e->pretest;
    ca.defer_exact = 0;
    ca.dyn_elsewhere = 0;
    ca.dbg = getenv("TOHC_BP_DBG") ? atoi(getenv("TOHC_BP_DBG")) : 0;
    // Who scans what (k_broadphase_unit's comment): when all a unit's box tests fit a few dozen rounds of one wavefront
    // (a small world; in "poses" mode every child moves and there is no static table), a wavefront per unit takes them
    // all and k_broadphase is not launched; else a wavefront per (unit, query child) walks the cluster table and the
    // moving children.  (Giving only the moving children of a large world to the unit kernel was measured too: no gain,
    // 0.72 vs 0.69 ms per launch on config 4.)
    const bool unitOk = e->unit_broadphase > 0 && cp.n_cast_sets >= e->unit_broadphase && nQ <= TOHC_BPU_MAX_Q && (unsigned)e->n_children < TOHC_BPU_MAX_CHILD && ca.dbg == 0;
    L.unitAll = unitOk && (size_t)nQ * ((dm.n_stat > 0 ? dm.n_stat_pad : 0) + dm.n_dyn_children) <= 64 * 96;
    // the exact stage of the pre-test in a kernel of its own, a lane per pair (k_exact)
    if (!L.unitAll) ca.defer_exact = (e->defer_exact && ca.pretest && TOHC_BP_EXACT && ca.dbg == 0) ? 1 : 0;
  }
  return TOHC_OK;
}
static int narrowBroadphase(tohc_engine *e, NarrowLaunch &L) {
  (void)e;
  if (L.oh.empty) return TOHC_OK;
  const CallParams &cp = L.cp;
  hipStream_t s = L.s;
  if (L.unitAll) {
    CullArgs cu = L.ca;
    cu.dyn_elsewhere = 1;   // (for this kernel: 1 = the static children are its own too)
    if (cp.cast) hipLaunchKernelGGL((k_broadphase_unit<true>), dim3((unsigned)cp.n_cast_sets), dim3(TOHC_BP_THREADS), 0, s, cu);
    else hipLaunchKernelGGL((k_broadphase_unit<false>), dim3((unsigned)cp.n_cast_sets), dim3(TOHC_BP_THREADS), 0, s, cu);
  } else if (L.ca.defer_exact) {
    if (cp.cast) hipLaunchKernelGGL((k_broadphase<true, true>), dim3((unsigned)L.blocks), dim3(TOHC_BP_THREADS), 0, s, L.ca);
    else hipLaunchKernelGGL((k_broadphase<false, true>), dim3((unsigned)L.blocks), dim3(TOHC_BP_THREADS), 0, s, L.ca);
  } else {
    if (cp.cast) hipLaunchKernelGGL((k_broadphase<true, false>), dim3((unsigned)L.blocks), dim3(TOHC_BP_THREADS), 0, s, L.ca);
    else hipLaunchKernelGGL((k_broadphase<false, false>), dim3((unsigned)L.blocks), dim3(TOHC_BP_THREADS), 0, s, L.ca);
  }
  return TOHC_OK;
}
// every open pass gets a tail of its own, in enqueue order (launchTail with several passes open makes ONE shared tail)
static int launchTailsEach(tohc_engine *e) {
  std::vector<tohc_engine::OpenHead> all;
  all.swap(e->open);
  int rc = TOHC_OK;
  for (const auto &h : all) {
    e->open.assign(1, h);
    const int r = launchTail(e);
    if (r != TOHC_OK && rc == TOHC_OK) rc = r;
  }
  return rc;
}
// holdTail: the pass stays open; the caller launches its tail (launchTailsEach / launchTail)
static int narrowFinish(tohc_engine *e, NarrowLaunch &L, bool holdTail = false) {
  const CallParams &cp = L.cp;
  hipStream_t s = L.s;
  hipEvent_t *ev = L.ev;
  tohc_engine::Lane &ln = e->lane[L.laneIdx];
  const NarrowArgs &a = L.oh.a;
  if (!L.oh.empty) {
    if (!L.unitAll && L.ca.defer_exact) {
      if (cp.cast) hipLaunchKernelGGL((k_exact<true>), dim3(TOHC_EXACT_GRID), dim3(64), 0, s, L.ca);
      else hipLaunchKernelGGL((k_exact<false>), dim3(TOHC_EXACT_GRID), dim3(64), 0, s, L.ca);
    }
    HIPCHK(recEv(e, ev, 3, s));
    const int gridN = 256 * 8;
    if (e->gjk_group == 1 && !L.gjk16) {  // one lane per pair, one wavefront per block
      if (cp.cast) hipLaunchKernelGGL((k_gjk<1, true, 64>), dim3(TOHC_GJK_GRID), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((k_gjk<1, false, 64>), dim3(TOHC_GJK_GRID), dim3(64), 0, s, a);
      if (cp.gjk_split > 0) {
        if (cp.cast) hipLaunchKernelGGL((k_gjk_cont<true>), dim3(TOHC_CONT_GRID), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_gjk_cont<false>), dim3(TOHC_CONT_GRID), dim3(64), 0, s, a);
      }
    } else if (e->gjk_group == 64) {
      if (cp.cast) hipLaunchKernelGGL((k_gjk<64, true, 256>), dim3(gridN), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_gjk<64, false, 256>), dim3(gridN), dim3(256), 0, s, a);
    } else {
      if (cp.cast) hipLaunchKernelGGL((k_gjk<16, true, 256>), dim3(gridN), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_gjk<16, false, 256>), dim3(gridN), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
  } else {
    HIPCHK(recEv(e, ev, 3, s));
  }
  HIPCHK(recEv(e, ev, 4, s));
  HIPCHK(hipEventRecord(ln.head_done, s));
  e->open.push_back(L.oh);
  tohc_engine::Pending pd;
  pd.slot = L.slot;
  pd.aabb_tests = countAabbTests(e, *L.h_qchildren, cp);
  pd.kind = cp.cast ? 1 : 0; pd.units = cp.n_cast_sets;
  e->pending.push_back(pd);
  e->sorted_valid = false;
  if (!L.append || L.noSlab) { e->last_robot_mode = L.robotMode; e->last_want_jac = L.wantJ; }
  if (!L.append) e->slab_state = 0;   // a new slab is being filled
  e->last_T = cp.T;
  if (!holdTail && (int)e->open.size() >= e->max_open) return launchTail(e);
  return TOHC_OK;
}
static int launchNarrow(tohc_engine *e, const DynMap &dm, const int *d_qchildren, const std::vector<int> &h_qchildren, CallParams cp,
                        bool robotMode, int slot, bool append) {
  NarrowLaunch L;
  int rc = narrowPrepare(e, dm, d_qchildren, h_qchildren, cp, robotMode, slot, append, L);
  if (rc != TOHC_OK) return rc;
  rc = narrowBroadphase(e, L);
  if (rc != TOHC_OK) return rc;
  return narrowFinish(e, L);
}

// ContinuousCheckTrajectory: one k_sweep launch writes the records itself (no pair list, no k_finish); always a new slab
static int launchSweep(tohc_engine *e, const DynMap &dm, const int *d_qchildren, const std::vector<int> &h_qchildren, CallParams cp, bool robotMode,
                       int slot) {
  const int nQ = (int)h_qchildren.size();
  {
    const int rc = launchTail(e);   // always a new slab: what is open belongs to the old one
    if (rc != TOHC_OK) return rc;
  }
  tohc_engine::Lane &ln = e->lane[slot % e->n_lanes];
  hipStream_t s = ln.stream;
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  HIPCHK(hipStreamWaitEvent(s, e->ev_fk, 0));
  for (auto &other : e->lane)
    if (&other != &ln) HIPCHK(hipStreamWaitEvent(s, other.done, 0));
  if (e->slab_read_pending) HIPCHK(hipStreamWaitEvent(s, e->ev_slab_read, 0));   // k_sweep writes records itself
  HIPCHK(hipMemsetAsync(e->d_cnt.p + 1 + slot, 0, sizeof(Counters), s));
  HIPCHK(hipMemsetAsync(e->d_nrec.p, 0, sizeof(unsigned int), s));
  cp.seg_cap = (unsigned int)(e->max_pairs / TOHC_NSHARD); cp.max_records = (unsigned int)e->max_records;
  cp.dbg_max_iter = 1000;
  cp.epa_small_v = e->epa_small_v; cp.epa_small_f = e->epa_small_f;
  HIPCHK(e->d_records.reserve(e->max_records));
  HIPCHK(recEv(e, ev, 2, s));
  if (nQ > 0 && cp.n_cast_sets > 0) {
    const size_t blocks = (size_t)cp.n_cast_sets * nQ;
    if (blocks > 0x7fffffffull) return fail(e, TOHC_E_CAPACITY, "sweep grid too large (%zu blocks)", blocks);
    // what the advancement may add to a resting geometry's reach: it is "integrated" with the rotation of
    // R * inverse(R), a few ulp away from identity, i.e. an angle of up to ~2e-3 rad about its origin
    float discMax = 0;
    for (int c = 0; c < e->n_children; ++c) {
      const int sh = e->h_child_shape[c];
      float n2 = 0;
      for (int k = 0; k < 3; ++k) {
        const float h = std::max(fabsf(e->h_shape_lmin[3 * sh + k]), fabsf(e->h_shape_lmax[3 * sh + k])) + 0.04f + e->d;
        n2 += h * h;
      }
      discMax = std::max(discMax, 2.0f * sqrtf(n2));
    }
    SweepArgs sa;
    NarrowArgs &a = sa.a;
    memset(&a, 0, sizeof a);
    a.sc = devScene(e); a.dm = dm; a.rb = devRobot(e); a.cp = cp;
    a.dynXf0 = e->d_dynXf.p; a.dynXf1 = e->d_dynXf.p;
    a.cnt = e->d_cnt.p + 1 + slot; a.n_records = e->d_nrec.p; a.rec_base = e->d_rec_base.p + slot; a.records = e->d_records.p;
    sa.qchildren = d_qchildren; sa.nQ = nQ; sa.dynAabb0 = e->d_dynAabb.p;
    sa.grow = 0.0011f + 4.0e-3f * discMax;
    hipLaunchKernelGGL(k_sweep, dim3((unsigned)blocks), dim3(64), 0, s, sa);
    HIPCHK(hipGetLastError());
  }
  for (int i = 3; i < NEV; ++i) HIPCHK(recEv(e, ev, i, s));   // the whole sweep is booked under "broadphase"
  HIPCHK(hipEventRecord(ln.done, s));
  tohc_engine::Pending pd;
  pd.slot = slot;
  pd.aabb_tests = 0;
  e->pending.push_back(pd);
  e->last_robot_mode = robotMode;
  e->last_want_jac = false;
  e->slab_state = 0;
  e->sorted_valid = false;
  e->last_T = cp.T;
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
  {
    const int rc = launchTail(e);
    if (rc != TOHC_OK) return rc;
  }
  for (auto &ln : e->lane) HIPCHK(hipStreamSynchronize(ln.stream));   // enqueues still reading the old tables
  HIPCHK(hipStreamSynchronize(e->tail_stream));
  e->fk_done = false;
  // through a pinned buffer: a copy from pageable memory is staged by the runtime and took 1.3 ms for the 0.9 MB of 64
  // problems -- with the GPU idle -- against 0.1 ms this way (scripts/probe_hinges.py)
  HIPCHK(e->h_dofs.reserve(n_sets * e->K));
  memcpy(e->h_dofs.p, dofs, n_sets * e->K * sizeof(double));
  HIPCHK(e->d_dofs.reserve(n_sets * e->K));
  HIPCHK(hipMemcpyAsync(e->d_dofs.p, e->h_dofs.p, n_sets * e->K * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e->d_linkQT.reserve(n_sets * e->n_links * 7));
  HIPCHK(e->d_linkZ.reserve(n_sets * e->n_links * 3));
  HIPCHK(e->d_objQT.reserve(n_sets * std::max(e->n_dyn_objR, 1) * 7));
  HIPCHK(e->d_dynXf.reserve(n_sets * std::max(e->n_dyn_childrenR, 1) * 12));
  HIPCHK(e->d_dynAabb.reserve(n_sets * std::max(e->n_dyn_childrenR, 1) * 6));
  HIPCHK(e->d_dynSph.reserve(n_sets * std::max(e->n_dyn_childrenR, 1)));
  e->staged_B = B; e->staged_T = T; e->staged = true;   // (the caller's buffer has been copied; FK follows on the same stream)
  return TOHC_OK;
}

// t_disc: the discrete pass covers the first t_disc timesteps of every staged trajectory (0 = all of them)
static int enqueueRobot(tohc_engine *e, int gap, int cast, int filter_mask, int want_jac, int append, int t_disc = 0) {
  if (!e) return TOHC_E_INVALID;
  if (!e->staged) return fail(e, TOHC_E_STATE, "stage the DOF vectors first (tohc_stage_dofs)");
  if (cast && gap < 1) return fail(e, TOHC_E_INVALID, "cast: gap must be >= 1");
  if (t_disc < 0 || t_disc > e->staged_T) return fail(e, TOHC_E_INVALID, "discrete window: 0 <= n_discrete <= staged timesteps");
  HIPCHK(hipSetDevice(e->device));
  int slot;
  int rc = nextSlot(e, &slot);
  if (rc != TOHC_OK) return rc;
  const int B = e->staged_B, T = e->staged_T;
  const size_t n_sets = (size_t)B * T;
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  DynMap dm = dynMapRobot(e);
  // FK once per linearization: every enqueue that starts a new slab runs it; appended passes reuse its tables
  if (!append || !e->fk_done) {
    rc = launchTail(e);   // (their k_finish kernels read the pose tables too)
    if (rc != TOHC_OK) return rc;
    hipStream_t s = e->stream;
    for (auto &ln : e->lane) HIPCHK(hipStreamWaitEvent(s, ln.done, 0));   // pending enqueues still read the tables
    HIPCHK(recEv(e, ev, 0, s));
    hipLaunchKernelGGL(k_fk, dim3((unsigned)n_sets), dim3(64), (size_t)e->n_links * 7 * sizeof(double), s, devRobot(e), devScene(e), dm, e->d_dofs.p,
                       (int)n_sets, e->d, e->d_linkQT.p, e->d_objQT.p, e->d_dynXf.p, e->d_dynAabb.p, e->d_linkZ.p, e->d_dynSph.p);
    HIPCHK(recEv(e, ev, 1, s));
    HIPCHK(hipEventRecord(e->ev_fk, s));
    e->fk_done = true;
  } else {
    hipStream_t ls = e->lane[slot % e->n_lanes].stream;
    HIPCHK(recEv(e, ev, 0, ls));
    HIPCHK(recEv(e, ev, 1, ls));
  }
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = T; cp.gap = cast ? gap : 0; cp.cast = cast; cp.filter_mask = filter_mask; cp.want_jac = want_jac; cp.d = e->d; cp.breaking = e->breaking;
  cp.t_disc = t_disc > 0 ? t_disc : T;
  const int units = cast ? B * std::max(T - gap, 0) : B * cp.t_disc;
  e->last_sets = (int64_t)n_sets;
  if (!append || (e->slab_state == 0 && e->pending.empty())) e->slab_B = B, e->slab_disc_passes = e->slab_cast_passes = 0, e->slab_gap = 0;
  if (cast) { if (e->slab_cast_passes++ > 0 && e->slab_gap != gap) e->slab_gap = -1; else e->slab_gap = gap; }
  else ++e->slab_disc_passes;
  // large enqueues are cut into unit ranges that run on different lanes: the kernels are latency bound, so
  // independent ranges overlap almost perfectly
  const int C = units >= 256 * e->chunks ? e->chunks : 1;
  for (int c = 0; c < C; ++c) {
    if (c > 0) {
      rc = nextSlot(e, &slot);
      if (rc != TOHC_OK) return rc;
      hipStream_t ls = e->lane[slot % e->n_lanes].stream;
      hipEvent_t *evc = &e->events[(size_t)slot * NEV];
      HIPCHK(recEv(e, evc, 0, ls));
      HIPCHK(recEv(e, evc, 1, ls));
    }
    cp.unit_base = (int)((int64_t)units * c / C);
    cp.n_cast_sets = (int)((int64_t)units * (c + 1) / C) - cp.unit_base;
    rc = launchNarrow(e, dm, e->d_qchildrenR.p, e->h_qchildrenR, cp, true, slot, append != 0 || c > 0);
    if (rc != TOHC_OK) return rc;
  }
  return TOHC_OK;
}

// tohc_enqueue_linearization: a new slab with the discrete pass (LinksVsAll, filter_mask) and the cast pass (CastVsAll,
// gap) of one SQP linearization -- what tohc_enqueue_links_vs_all(append = 0) followed by tohc_enqueue_cast_vs_all(append
// = 1) enqueue, same records, same counters -- with ONE broadphase launch for both passes when the engine's set-up allows
// it (k_broadphase_both), else as those two calls.
// t_disc: 0 (all), or the number of timesteps the discrete pass covers: the first t_disc of every staged trajectory; the
// rest are only there as second poses of this shard's last cast pairs (tohc_enqueue_linearization_window)
static int enqueueBoth(tohc_engine *e, int filter_mask, int gap, int want_jac, int t_disc = 0) {
  if (!e) return TOHC_E_INVALID;
  if (!e->staged) return fail(e, TOHC_E_STATE, "stage the DOF vectors first (tohc_stage_dofs)");
  if (gap < 1) return fail(e, TOHC_E_INVALID, "cast: gap must be >= 1");
  const int B = e->staged_B, T = e->staged_T;
  if (t_disc == 0) t_disc = T;
  if (t_disc < 1 || t_disc > T) return fail(e, TOHC_E_INVALID, "linearization window: 1 <= n_discrete <= staged timesteps");
  // one launch covers the discrete pass's units: all T timesteps, or the first T - gap (then exactly the cast pass's units)
  const bool fuse = e->fuse_broadphase && e->n_lanes >= 2 && e->chunks == 1 && e->defer_exact && e->pretest && T > gap && !getenv("TOHC_BP_DBG") &&
                    (t_disc == T || t_disc == T - gap);
  if (!fuse) {
    const int rc = enqueueRobot(e, 0, 0, filter_mask, want_jac, 0, t_disc);
    return rc != TOHC_OK ? rc : enqueueRobot(e, gap, 1, -1, want_jac, 1);
  }
  HIPCHK(hipSetDevice(e->device));
  if ((int)e->pending.size() + 2 > RING) {
    const int rc = syncEngine(e);
    if (rc != TOHC_OK) return rc;
  }
  const int slotD = (int)e->pending.size(), slotC = slotD + 1;
  const size_t n_sets = (size_t)B * T;
  DynMap dm = dynMapRobot(e);
  int rc = launchTail(e);   // (their k_finish kernels read the pose tables too)
  if (rc != TOHC_OK) return rc;
  {   // FK, as every enqueue that starts a slab
    hipEvent_t *ev = &e->events[(size_t)slotD * NEV];
    hipStream_t s = e->stream;
    for (auto &ln : e->lane) HIPCHK(hipStreamWaitEvent(s, ln.done, 0));   // pending enqueues still read the tables
    HIPCHK(recEv(e, ev, 0, s));
    hipLaunchKernelGGL(k_fk, dim3((unsigned)n_sets), dim3(64), (size_t)e->n_links * 7 * sizeof(double), s, devRobot(e), devScene(e), dm, e->d_dofs.p,
                       (int)n_sets, e->d, e->d_linkQT.p, e->d_objQT.p, e->d_dynXf.p, e->d_dynAabb.p, e->d_linkZ.p, e->d_dynSph.p);
    HIPCHK(recEv(e, ev, 1, s));
    HIPCHK(hipEventRecord(e->ev_fk, s));
    e->fk_done = true;
  }
  CallParams cpD, cpC;
  memset(&cpD, 0, sizeof cpD);
  cpD.T = T; cpD.gap = 0; cpD.cast = 0; cpD.filter_mask = filter_mask; cpD.want_jac = want_jac; cpD.d = e->d; cpD.breaking = e->breaking;
  cpD.t_disc = t_disc;
  cpD.unit_base = 0; cpD.n_cast_sets = B * t_disc;
  cpC = cpD;
  cpC.gap = gap; cpC.cast = 1; cpC.filter_mask = -1; cpC.n_cast_sets = B * (T - gap);
  e->last_sets = (int64_t)n_sets;
  e->slab_B = B; e->slab_disc_passes = 1; e->slab_cast_passes = 1; e->slab_gap = gap;
  NarrowLaunch LD, LC;
  rc = narrowPrepare(e, dm, e->d_qchildrenR.p, e->h_qchildrenR, cpD, true, slotD, false, LD);
  if (rc != TOHC_OK) return rc;
  // (what the first pass's narrowFinish will say about the slab, already now: the second pass is prepared before it)
  e->slab_state = 0; e->last_robot_mode = true; e->last_want_jac = LD.wantJ;
  {
    hipStream_t ls = e->lane[slotC % e->n_lanes].stream;
    hipEvent_t *evc = &e->events[(size_t)slotC * NEV];
    HIPCHK(recEv(e, evc, 0, ls));
    HIPCHK(recEv(e, evc, 1, ls));
  }
  rc = narrowPrepare(e, dm, e->d_qchildrenR.p, e->h_qchildrenR, cpC, true, slotC, true, LC);
  if (rc != TOHC_OK) {   // the first pass alone, so that the engine's books stay whole
    (void)narrowBroadphase(e, LD);
    (void)narrowFinish(e, LD);
    return rc;
  }
  const bool both = !LD.oh.empty && !LC.oh.empty && !LD.unitAll && !LC.unitAll && LD.ca.defer_exact && LC.ca.defer_exact && LD.laneIdx != LC.laneIdx &&
                    LD.nQ == LC.nQ;
  if (both) {
    BothArgs ba;
    ba.d = LD.ca; ba.c = LC.ca;
    HIPCHK(hipEventRecord(e->ev_prep, LD.s));          // the first pass's counters are cleared
    HIPCHK(hipStreamWaitEvent(LC.s, e->ev_prep, 0));
    hipLaunchKernelGGL(k_broadphase_both, dim3((unsigned)LD.blocks), dim3(TOHC_BP_THREADS), 0, LC.s, ba);
    HIPCHK(hipEventRecord(e->ev_fused, LC.s));
    HIPCHK(hipStreamWaitEvent(LD.s, e->ev_fused, 0));
    HIPCHK(recEv(e, LD.ev, 2, LD.s));   // the shared launch is booked on the second pass only (cum_ms_broadphase adds the passes up)
    const bool headsFirst = e->heads_first && e->max_open == 1;
    rc = narrowFinish(e, LD, headsFirst);
    if (rc != TOHC_OK) return rc;
    rc = narrowFinish(e, LC, headsFirst);
    if (rc != TOHC_OK || !headsFirst) return rc;
    return launchTailsEach(e);
  }
  rc = narrowBroadphase(e, LD);
  if (rc == TOHC_OK) rc = narrowFinish(e, LD);
  if (rc != TOHC_OK) return rc;
  rc = narrowBroadphase(e, LC);
  return rc != TOHC_OK ? rc : narrowFinish(e, LC);
}
int tohc_enqueue_linearization(tohc_engine *e, int filter_mask, int gap, int want_jacobians) {
  return enqueueBoth(e, filter_mask, gap, want_jacobians);
}
int tohc_enqueue_linearization_window(tohc_engine *e, int filter_mask, int gap, int want_jacobians, int n_discrete) {
  if (e && n_discrete <= 0) return fail(e, TOHC_E_INVALID, "linearization window: 1 <= n_discrete <= staged timesteps");
  return enqueueBoth(e, filter_mask, gap, want_jacobians, n_discrete);
}

int tohc_enqueue_links_vs_all(tohc_engine *e, int filter_mask, int want_jacobians, int append) {
  return enqueueRobot(e, 0, 0, filter_mask, want_jacobians, append);
}
int tohc_enqueue_cast_vs_all(tohc_engine *e, int gap, int want_jacobians, int append) {
  return enqueueRobot(e, gap, 1, -1, want_jacobians, append);
}

int tohc_set_kernel_timing(tohc_engine *e, int on) {
  if (!e) return TOHC_E_INVALID;
  if (!e->pending.empty() || !e->open.empty()) {   // (events of enqueues in flight are read at their sync)
    HIPCHK(hipSetDevice(e->device));
    const int rc = syncEngine(e);
    if (rc != TOHC_OK) return rc;
  }
  e->timing = on != 0;
  return TOHC_OK;
}

int tohc_sync(tohc_engine *e) {
  if (!e) return TOHC_E_INVALID;
  HIPCHK(hipSetDevice(e->device));
  return syncEngine(e);
}

// sync, check that there is a slab to hand out, and -- where the 64-bit key can hold the scene -- put it in canonical
// order ON THE DEVICE (d_sorted / d_gsorted).  *n = records, *devSorted = whether the device copies are in order.
static int prepareSlab(tohc_engine *e, size_t *n_out, bool *devSorted) {
  *n_out = 0;
  *devSorted = false;
  HIPCHK(hipSetDevice(e->device));
  int rc = syncEngine(e);
  if (rc != TOHC_OK) return rc;
  if (e->slab_state < 0) return fail(e, TOHC_E_STATE, "fetch: the last sync of this slab failed (capacity or HIP error); there is no valid slab");
  if (e->slab_state == 0) return TOHC_OK;   // nothing enqueued yet: an empty slab
  const size_t n = std::min<size_t>(*e->h_nrec.p, e->d_records.cap);
  *n_out = n;
  if (n == 0) return TOHC_OK;
  const int K = e->last_robot_mode ? e->K : 1;
  const bool haveG = e->last_want_jac;
  // canonical order: Bullet's own order follows DBVT traversal and is history dependent.
  // discrete records (cctype none) sort before cast records of the same (problem, t) when both were appended.
  const bool devSort = !e->host_sort && e->last_sets <= SORT_MAX_SETS && e->n_children <= SORT_MAX_CHILDREN && e->children_in_object_order;
  if (!devSort) return TOHC_OK;
  if (e->sorted_valid) { *devSorted = true; return TOHC_OK; }   // this slab has been ordered already (fetch after fetch_device)
  size_t tmpBytes = 0;
  HIPCHK(sortTempBytes(n, &tmpBytes));
  HIPCHK(e->d_sort_tmp.reserve(tmpBytes)); HIPCHK(e->d_sort_keys.reserve(2 * n)); HIPCHK(e->d_sort_idx.reserve(2 * n));
  HIPCHK(e->d_sorted.reserve(n));
  if (haveG) HIPCHK(e->d_gsorted.reserve(n * 2 * K));
  HIPCHK(sortRecords(e->d_records.p, haveG ? e->d_grads.p : nullptr, n, e->last_T, K, e->d_obj_off.p, e->d_sort_keys.p, e->d_sort_idx.p, e->d_sort_tmp.p, tmpBytes,
                     e->d_sorted.p, haveG ? e->d_gsorted.p : nullptr, e->stream));
  e->sorted_valid = true;
  *devSorted = true;
  return TOHC_OK;
}

int tohc_fetch(tohc_engine *e, const tohc_record **records, const double **grads, int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (records) *records = nullptr;
  if (grads) *grads = nullptr;
  size_t n = 0;
  bool devSort = false;
  const bool trace = getenv("TOHC_TRACE_FETCH") != nullptr;
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_a = now();
  int rc = prepareSlab(e, &n, &devSort);
  if (rc != TOHC_OK || n == 0) return rc;
  hipStream_t s = e->stream;
  const int K = e->last_robot_mode ? e->K : 1;
  const bool haveG = e->last_want_jac;
  HIPCHK(e->h_sorted.reserve(n));
  if (haveG) HIPCHK(e->h_gsorted.reserve(n * 2 * K));
  if (devSort) {
    if (trace) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[fetch] device sort of %zu records: %.3f ms\n", n, now() - t_a); }
    HIPCHK(hipMemcpyAsync(e->h_sorted.p, e->d_sorted.p, n * sizeof(tohc_record), hipMemcpyDeviceToHost, s));
    if (haveG) HIPCHK(hipMemcpyAsync(e->h_gsorted.p, e->d_gsorted.p, n * 2 * K * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  } else {
    HIPCHK(e->h_records.reserve(n));
    HIPCHK(hipMemcpyAsync(e->h_records.p, e->d_records.p, n * sizeof(tohc_record), hipMemcpyDeviceToHost, s));
    if (haveG) {
      HIPCHK(e->h_grads.reserve(n * 2 * K));
      HIPCHK(hipMemcpyAsync(e->h_grads.p, e->d_grads.p, n * 2 * K * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    const tohc_record *r = e->h_records.p;
    std::stable_sort(idx.begin(), idx.end(), [r](uint32_t i, uint32_t j) {  // stable, like the device radix sort: equal keys keep slab order
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
  }
  if (trace)
    fprintf(stderr, "[fetch] %s path (host_sort=%d sets=%lld children=%d), %zu records, K=%d: %.3f ms\n", devSort ? "device-sort" : "host-sort",
            (int)e->host_sort, (long long)e->last_sets, e->n_children, n, K, now() - t_a);
  *n_records = (int)n;
  if (records) *records = e->h_sorted.p;
  if (grads) *grads = haveG ? e->h_gsorted.p : nullptr;
  return TOHC_OK;
}

int tohc_fetch_device(tohc_engine *e, const tohc_record **d_records, const double **d_grads, int *n_records, int *row_doubles) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records || !d_records) return fail(e, TOHC_E_INVALID, "null result pointer");
  *n_records = 0;
  *d_records = nullptr;
  if (d_grads) *d_grads = nullptr;
  if (row_doubles) *row_doubles = 0;
  size_t n = 0;
  bool devSort = false;
  int rc = prepareSlab(e, &n, &devSort);
  if (rc != TOHC_OK || n == 0) return rc;
  if (!devSort) return fail(e, TOHC_E_UNSUPPORTED, "fetch_device: this slab can only be ordered on the host (more than %d pose sets or %d geometries); use tohc_fetch", SORT_MAX_SETS, SORT_MAX_CHILDREN);
  HIPCHK(hipStreamSynchronize(e->stream));   // the caller may read the device slab on any stream
  const int K = e->last_robot_mode ? e->K : 1;
  *n_records = (int)n;
  *d_records = e->d_sorted.p;
  if (d_grads) *d_grads = e->last_want_jac ? e->d_gsorted.p : nullptr;
  if (row_doubles) *row_doubles = e->last_want_jac ? 2 * K : 0;
  return TOHC_OK;
}

int tohc_fetch_hinges_begin(tohc_engine *e, double dist_pen_discrete, double dist_pen_cast) {
  if (!e) return TOHC_E_INVALID;
  if (e->hinge_open) return fail(e, TOHC_E_STATE, "fetch_hinges_begin: the previous begin has not been ended");
  HIPCHK(hipSetDevice(e->device));
  int rc = syncEngine(e);
  if (rc != TOHC_OK) return rc;
  if (e->slab_state != 1) return fail(e, TOHC_E_STATE, e->slab_state < 0 ? "fetch_hinges: the last sync of this slab failed; there is no valid slab" : "fetch_hinges: nothing has been enqueued");
  if (!e->last_robot_mode || !e->last_want_jac) return fail(e, TOHC_E_STATE, "fetch_hinges: the slab holds no gradient rows (enqueue with want_jacobians on staged DOF vectors)");
  if (e->slab_disc_passes > 1 || e->slab_cast_passes > 1 || e->slab_gap < 0)
    return fail(e, TOHC_E_UNSUPPORTED, "fetch_hinges: the slab holds %d discrete and %d cast passes; one of each at most", e->slab_disc_passes, e->slab_cast_passes);
  const int B = e->slab_B, T = e->last_T, K = e->K, gap = e->slab_cast_passes ? e->slab_gap : 0;
  if (e->last_sets > SORT_MAX_SETS || e->n_children > SORT_MAX_CHILDREN || !e->children_in_object_order)
    return fail(e, TOHC_E_UNSUPPORTED, "fetch_hinges: more than %d pose sets or %d geometries", SORT_MAX_SETS, SORT_MAX_CHILDREN);
  const size_t n = std::min<size_t>(*e->h_nrec.p, e->d_records.cap);
  const size_t ncols = (size_t)B * T * K, nseg = (size_t)B * T * 2;
  if (ncols + 1 > 0x7fffffffull) return fail(e, TOHC_E_CAPACITY, "fetch_hinges: %zu trajectory variables", ncols);
  hipStream_t s = e->hinge_stream;
  tohc_engine::HingeHost &h = e->hh[e->hinge_cur];
  h.n = n; h.B = B; h.T = T; h.K = K; h.gap = gap;
  size_t tmpBytes = 0;
  HIPCHK(sortTempBytes(std::max<size_t>(n, 1), &tmpBytes));
  HIPCHK(e->d_htmp.reserve(tmpBytes)); HIPCHK(e->d_hkeys.reserve(2 * n + 2)); HIPCHK(e->d_hidx.reserve(2 * n + 2));
  HIPCHK(e->d_hrhs.reserve(n)); HIPCHK(e->d_hdist.reserve(n)); HIPCHK(e->d_hwa.reserve(n)); HIPCHK(e->d_hwb.reserve(n)); HIPCHK(e->d_hrowt.reserve(n));
  HIPCHK(e->d_hseg.reserve(nseg + 1)); HIPCHK(e->d_hrowoff.reserve((size_t)B + 1)); HIPCHK(e->d_hcnt.reserve(ncols)); HIPCHK(e->d_hcoloff.reserve(ncols + 1));
  HIPCHK(e->d_hcolidx.reserve(n * 2 * (size_t)K)); HIPCHK(e->d_hval.reserve(n * 2 * (size_t)K));   // the bound: every coefficient non-zero
  HIPCHK(h.rhs.reserve(n)); HIPCHK(h.dist.reserve(n)); HIPCHK(h.row_t.reserve(n)); HIPCHK(h.row_off.reserve((size_t)B + 1)); HIPCHK(h.col_off.reserve(ncols + 1));
  HIPCHK(sortKeys(e->d_records.p, n, T, e->d_obj_off.p, e->d_hkeys.p, e->d_hidx.p, e->d_htmp.p, tmpBytes, s));
  HingeArgs a;
  a.rec = e->d_records.p; a.grads = e->d_grads.p; a.idx = e->d_hidx.p + n; a.keys = e->d_hkeys.p + n; a.n = (unsigned int)n;
  a.B = B; a.T = T; a.K = K; a.gap = gap; a.pen_disc = dist_pen_discrete; a.pen_cast = dist_pen_cast;
  a.rhs = e->d_hrhs.p; a.dist = e->d_hdist.p; a.wa = e->d_hwa.p; a.wb = e->d_hwb.p; a.row_t = e->d_hrowt.p; a.seg = e->d_hseg.p; a.row_off = e->d_hrowoff.p;
  a.colcnt = e->d_hcnt.p; a.col_off = e->d_hcoloff.p; a.col_idx = e->d_hcolidx.p; a.col_val = e->d_hval.p;
  if (n > 0) hipLaunchKernelGGL(k_hinge_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_hinge_seg, dim3((unsigned)((nseg + 1 + 255) / 256)), dim3(256), 0, s, a);
  const bool wide = K > TOHC_HINGE_MAXK;
  if (wide) hipLaunchKernelGGL((k_hinge_cols_wide<false>), dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_hinge_cols<false>), dim3((unsigned)(B * T)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_hinge_scan, dim3(1), dim3(1024), 0, s, (const unsigned int *)e->d_hcnt.p, (unsigned int)ncols, e->d_hcoloff.p);
  if (wide) hipLaunchKernelGGL((k_hinge_cols_wide<true>), dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_hinge_cols<true>), dim3((unsigned)(B * T)), dim3(256), 0, s, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ev_slab_read, s));
  e->slab_read_pending = true;
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(h.rhs.p, e->d_hrhs.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h.dist.p, e->d_hdist.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h.row_t.p, e->d_hrowt.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipMemcpyAsync(h.row_off.p, e->d_hrowoff.p, ((size_t)B + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h.col_off.p, e->d_hcoloff.p, (ncols + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(e->ev_hinge_small, s));
  e->hinge_open = true;
  return TOHC_OK;
}

int tohc_fetch_hinges_end(tohc_engine *e, tohc_hinges *out) {
  if (!e) return TOHC_E_INVALID;
  if (!out) return fail(e, TOHC_E_INVALID, "fetch_hinges_end: null result pointer");
  memset(out, 0, sizeof *out);
  if (!e->hinge_open) return fail(e, TOHC_E_STATE, "fetch_hinges_end: no begin is outstanding");
  HIPCHK(hipSetDevice(e->device));
  e->hinge_open = false;
  tohc_engine::HingeHost &h = e->hh[e->hinge_cur];
  e->hinge_cur ^= 1;
  HIPCHK(hipEventSynchronize(e->ev_hinge_small));
  const size_t ncols = (size_t)h.B * h.T * h.K;
  const size_t nnz = (size_t)h.col_off.p[ncols];
  if (nnz > e->d_hval.cap) return fail(e, TOHC_E_HIP, "fetch_hinges_end: inconsistent column offsets");
  HIPCHK(h.idx.reserve(nnz)); HIPCHK(h.val.reserve(nnz));
  if (nnz > 0) {   // the bulk of the bytes: called after the next linearization was enqueued, this runs beside its kernels
    HIPCHK(hipMemcpyAsync(h.idx.p, e->d_hcolidx.p, nnz * sizeof(int32_t), hipMemcpyDeviceToHost, e->hinge_stream));
    HIPCHK(hipMemcpyAsync(h.val.p, e->d_hval.p, nnz * sizeof(double), hipMemcpyDeviceToHost, e->hinge_stream));
  }
  HIPCHK(hipStreamSynchronize(e->hinge_stream));
  out->n_problems = h.B; out->T = h.T; out->K = h.K; out->gap = h.gap;
  out->n_rows = (int64_t)h.n; out->nnz = (int64_t)nnz;
  out->row_off = reinterpret_cast<const int64_t *>(h.row_off.p); out->rhs = h.rhs.p; out->dist = h.dist.p; out->row_t = h.row_t.p;
  out->col_off = reinterpret_cast<const int64_t *>(h.col_off.p); out->col_idx = h.idx.p; out->col_val = h.val.p;
  return TOHC_OK;
}

int tohc_fetch_hinges(tohc_engine *e, double dist_pen_discrete, double dist_pen_cast, tohc_hinges *out) {
  int rc = tohc_fetch_hinges_begin(e, dist_pen_discrete, dist_pen_cast);
  if (rc == TOHC_OK) rc = tohc_fetch_hinges_end(e, out);
  return rc;
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

// all_pairs: AllVsAll -- every unordered object pair once (the broadphase keeps X > Q), contacts kept up to the breaking
// threshold 2.001 d instead of d (boxes and the pre-test's bound grown to match), the manifold round trip applied to the
// fetched records
static int posesQuery(tohc_engine *e, const double *obj_T, const double *cast_T1, int n_sets, const int32_t *query, int n_query, int filter_mask,
                      int cast, const tohc_record **records, int *n_records, bool all_pairs = false) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  if (!obj_T || n_sets < 1 || n_query < 0 || (n_query > 0 && !query) || (cast && !cast_T1)) return fail(e, TOHC_E_INVALID, "null pose or query input");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
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
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  HIPCHK(recEv(e, ev, 0, s));
  HIPCHK(recEv(e, ev, 1, s));
  const size_t nobj = e->n_objects, nch = e->n_children;
  HIPCHK(upload(e->d_objQT, obj_T, (size_t)n_sets * nobj * 7, s));
  e->h_qinfoP = queryInfo(e, qch, e->h_child_dynP);
  HIPCHK(upload(e->d_qchildrenP, e->h_qinfoP, s));
  HIPCHK(e->d_dynXf.reserve((size_t)n_sets * nch * 12)); HIPCHK(e->d_dynAabb.reserve((size_t)n_sets * nch * 6));
  HIPCHK(e->d_dynSph.reserve((size_t)n_sets * nch));
  DynMap dm = dynMapPoses(e);
  const size_t tot = (size_t)n_sets * nch;
  const float keep = all_pairs ? e->breaking : e->d;
  hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_objQT.p, n_sets,
                     keep, e->d_dynXf.p, e->d_dynAabb.p, e->d_dynSph.p);
  std::vector<double> T1;
  if (cast) {
    // end-of-sweep poses: the query links move, everything else keeps its pose
    T1.assign(obj_T, obj_T + (size_t)n_sets * nobj * 7);
    for (int st = 0; st < n_sets; ++st)
      for (int i = 0; i < n_query; ++i) memcpy(&T1[((size_t)st * nobj + query[i]) * 7], cast_T1 + ((size_t)st * n_query + i) * 7, 7 * sizeof(double));
    HIPCHK(upload(e->d_objQT1, T1, s));
    HIPCHK(e->d_dynXf1.reserve((size_t)n_sets * nch * 12)); HIPCHK(e->d_dynAabb1.reserve((size_t)n_sets * nch * 6));
    hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_objQT1.p,
                       n_sets, e->d, e->d_dynXf1.p, e->d_dynAabb1.p);
  }
  HIPCHK(hipEventRecord(e->ev_fk, s));
  e->fk_done = false;
  HIPCHK(hipStreamSynchronize(s));  // caller-owned / local host buffers may go away
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = 1; cp.gap = 0; cp.cast = cast; cp.filter_mask = filter_mask; cp.want_jac = 0; cp.d = keep; cp.breaking = e->breaking;
  cp.ordered_pairs = all_pairs ? 1 : 0;
  cp.n_cast_sets = n_sets;
  e->last_sets = n_sets;
  rc = launchNarrow(e, dm, e->d_qchildrenP.p, qch, cp, false, slot, false);
  if (rc != TOHC_OK) return rc;
  const double *g = nullptr;
  return tohc_fetch(e, records, &g, n_records);
}

// AllVsAll (bullet_collision_checker.cpp:463-498) reads its contacts from Bullet's persistent manifolds, one per convex
// pair, each fed one point by GJK: btManifoldResult::addContactPoint (btManifoldResult.cpp:85-126) stores the point in the
// two OBJECTS' frames and refreshContactPoints (btPersistentManifold.cpp:199-259; own manifold,
// btConvexConvexAlgorithm.cpp:583-586) derives the world points back from those and the distance as (posA - posB) . n.
// Same fp32 steps here, on the fetched records (a few hundred of them: host code).
int tohc_all_vs_all_poses(tohc_engine *e, const double *obj_T, const tohc_record **records, int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  std::vector<int32_t> query((size_t)e->n_objects);
  std::iota(query.begin(), query.end(), 0);
  const tohc_record *rec = nullptr;
  int n = 0;
  int rc = posesQuery(e, obj_T, nullptr, 1, query.data(), (int)query.size(), -1, 0, &rec, &n, true);
  if (rc != TOHC_OK) return rc;
  tohc_record *out = const_cast<tohc_record *>(rec);   // the engine's own host buffer
  int m = 0;
  for (int i = 0; i < n; ++i) {
    tohc_record r = rec[i];
    // the manifold walk asks CanCollide again in the manifold's order of the two objects (:484); the broadphase asked at
    // (lower, higher) object number like the near callback (:418) -- with one-directional exclusions both count
    if (!e->h_allowed.empty() && !e->h_allowed[(size_t)r.idA * e->n_objects + r.idB]) continue;
    const Xf tA = xfFrom7(obj_T + 7 * (size_t)r.idA), tB = xfFrom7(obj_T + 7 * (size_t)r.idB);
    const V3 nW = mk((float)r.n[0], (float)r.n[1], (float)r.n[2]);
    const V3 pointA = mk((float)r.ptA[0], (float)r.ptA[1], (float)r.ptA[2]), pointB = mk((float)r.ptB[0], (float)r.ptB[1], (float)r.ptB[2]);
    const V3 localA = mulVM(pointA - tA.o, tA.b), localB = mulVM(pointB - tB.o, tB.b);   // btTransform::invXform
    const V3 posA = xfApply(tA, localA), posB = xfApply(tB, localB);
    const float dist = dot(posA - posB, nW);
    if (!(dist <= e->breaking)) continue;   // validContactDistance
    const V3 diff = posB - (posA - nW * dist);
    if (dot(diff, diff) > e->breaking * e->breaking) continue;
    r.dist = dist;
    r.ptA[0] = posA.x, r.ptA[1] = posA.y, r.ptA[2] = posA.z;
    r.ptB[0] = posB.x, r.ptB[1] = posB.y, r.ptB[2] = posB.z;
    out[m++] = r;
  }
  if (records) *records = rec;
  *n_records = m;
  return TOHC_OK;
}

int tohc_links_vs_all_poses(tohc_engine *e, const double *obj_T, int n_sets, const int32_t *query, int n_query, int filter_mask,
                            const tohc_record **records, int *n_records) {
  return posesQuery(e, obj_T, nullptr, n_sets, query, n_query, filter_mask, 0, records, n_records);
}
int tohc_cast_vs_all_poses(tohc_engine *e, const double *obj_T, const double *cast_T1, int n_sets, const int32_t *query, int n_query,
                           const tohc_record **records, int *n_records) {
  return posesQuery(e, obj_T, cast_T1, n_sets, query, n_query, -1, 1, records, n_records);
}

int tohc_continuous_check_trajectory(tohc_engine *e, const double *dofs, int B, int T, const tohc_record **records, int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (records) *records = nullptr;
  int rc = tohc_stage_dofs(e, dofs, B, T);
  if (rc != TOHC_OK) return rc;
  if (T < 2) return TOHC_OK;   // no segment
  int slot;
  rc = nextSlot(e, &slot);
  if (rc != TOHC_OK) return rc;
  const size_t n_sets = (size_t)B * T;
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  DynMap dm = dynMapRobot(e);
  hipStream_t s = e->stream;
  for (auto &ln : e->lane) HIPCHK(hipStreamWaitEvent(s, ln.done, 0));
  HIPCHK(recEv(e, ev, 0, s));
  hipLaunchKernelGGL(k_fk, dim3((unsigned)n_sets), dim3(64), (size_t)e->n_links * 7 * sizeof(double), s, devRobot(e), devScene(e), dm, e->d_dofs.p,
                     (int)n_sets, e->d, e->d_linkQT.p, e->d_objQT.p, e->d_dynXf.p, e->d_dynAabb.p);
  HIPCHK(recEv(e, ev, 1, s));
  HIPCHK(hipEventRecord(e->ev_fk, s));
  e->fk_done = true;
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = T; cp.gap = 1; cp.cast = 1; cp.filter_mask = TOHC_GROUP_KINBODY; cp.d = e->d; cp.breaking = e->breaking;
  cp.unit_base = 0; cp.n_cast_sets = B * (T - 1);
  e->last_sets = (int64_t)n_sets;
  rc = launchSweep(e, dm, e->d_qchildrenR.p, e->h_qchildrenR, cp, true, slot);
  if (rc != TOHC_OK) return rc;
  const double *g = nullptr;
  return tohc_fetch(e, records, &g, n_records);
}

int tohc_continuous_check_poses(tohc_engine *e, const double *obj_T, int n_steps, const int32_t *query, int n_query, const tohc_record **records,
                                int *n_records) {
  if (!e) return TOHC_E_INVALID;
  if (!n_records) return fail(e, TOHC_E_INVALID, "n_records is null");
  *n_records = 0;
  if (records) *records = nullptr;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  if (!obj_T || n_steps < 1 || n_query < 0 || (n_query > 0 && !query)) return fail(e, TOHC_E_INVALID, "null pose or query input");
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(quiesce(e));
  std::vector<int> qch;
  for (int i = 0; i < n_query; ++i) {
    int o = query[i];
    if (o < 0 || o >= e->n_objects) return fail(e, TOHC_E_INVALID, "query object %d out of range", o);
    for (int k = 0; k < e->h_obj_cnt[o]; ++k) qch.push_back(e->h_obj_off[o] + k);
  }
  if (n_steps < 2) return TOHC_OK;
  int slot;
  int rc = nextSlot(e, &slot);
  if (rc != TOHC_OK) return rc;
  e->staged = false;
  hipStream_t s = e->stream;
  hipEvent_t *ev = &e->events[(size_t)slot * NEV];
  HIPCHK(recEv(e, ev, 0, s));
  HIPCHK(recEv(e, ev, 1, s));
  const size_t nobj = e->n_objects, nch = e->n_children;
  HIPCHK(upload(e->d_objQT, obj_T, (size_t)n_steps * nobj * 7, s));
  e->h_qinfoP = queryInfo(e, qch, e->h_child_dynP);
  HIPCHK(upload(e->d_qchildrenP, e->h_qinfoP, s));
  HIPCHK(e->d_dynXf.reserve((size_t)n_steps * nch * 12)); HIPCHK(e->d_dynAabb.reserve((size_t)n_steps * nch * 6));
  DynMap dm = dynMapPoses(e);
  const size_t tot = (size_t)n_steps * nch;
  hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_objQT.p, n_steps, e->d, e->d_dynXf.p,
                     e->d_dynAabb.p);
  HIPCHK(hipEventRecord(e->ev_fk, s));
  e->fk_done = false;
  HIPCHK(hipStreamSynchronize(s));
  CallParams cp;
  memset(&cp, 0, sizeof cp);
  cp.T = n_steps; cp.gap = 1; cp.cast = 1; cp.filter_mask = TOHC_GROUP_KINBODY; cp.d = e->d; cp.breaking = e->breaking;
  cp.unit_base = 0; cp.n_cast_sets = n_steps - 1;
  e->last_sets = n_steps;
  rc = launchSweep(e, dm, e->d_qchildrenP.p, qch, cp, false, slot);
  if (rc != TOHC_OK) return rc;
  const double *g = nullptr;
  return tohc_fetch(e, records, &g, n_records);
}

// RayCastCollision: pose tables of ONE set are on the device (k_fk or k_dyn_children, stream e->stream); one k_ray
// launch, results straight into the caller's array
static int launchRays(tohc_engine *e, const DynMap &dm, const double *from, const double *to, int n_rays, float *closest) {
  hipStream_t s = e->stream;
  std::vector<float> rays((size_t)n_rays * 8, 0.0f);
  for (int i = 0; i < n_rays; ++i)
    for (int k = 0; k < 3; ++k) {   // toBt(): double -> float (bullet_collision_checker.cpp:72-74)
      rays[(size_t)8 * i + k] = (float)from[3 * i + k];
      rays[(size_t)8 * i + 4 + k] = (float)to[3 * i + k];
    }
  std::vector<unsigned int> init((size_t)n_rays, 0x3f800000u);   // 1.0f = ClosestRayResultCallback's initial m_closestHitFraction
  HIPCHK(upload(e->d_rays, rays, s));
  HIPCHK(upload(e->d_ray_closest, init, s));
  const size_t tot = (size_t)n_rays * e->n_children;
  if (tot > 0) {
    if ((tot + 255) / 256 > 0x7fffffffull) return fail(e, TOHC_E_CAPACITY, "ray grid too large (%zu ray x child tests)", tot);
    RayArgs ra;
    ra.sc = devScene(e); ra.dm = dm; ra.dynXf = e->d_dynXf.p; ra.rays = e->d_rays.p; ra.closest = e->d_ray_closest.p; ra.n_rays = n_rays;
    hipLaunchKernelGGL(k_ray, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ra);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(closest, e->d_ray_closest.p, (size_t)n_rays * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return TOHC_OK;
}

int tohc_ray_cast_poses(tohc_engine *e, const double *obj_T, const double *from, const double *to, int n_rays, float *closest_fraction) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  if (!obj_T || n_rays < 0 || (n_rays > 0 && (!from || !to || !closest_fraction))) return fail(e, TOHC_E_INVALID, "null pose, ray or result pointer");
  if (n_rays == 0) return TOHC_OK;
  HIPCHK(hipSetDevice(e->device));
  int rc = syncEngine(e);
  if (rc != TOHC_OK) return rc;
  HIPCHK(quiesce(e));
  e->staged = false;   // the pose tables below overwrite the staged robot tables
  hipStream_t s = e->stream;
  const size_t nobj = e->n_objects, nch = e->n_children;
  HIPCHK(upload(e->d_objQT, obj_T, nobj * 7, s));
  HIPCHK(e->d_dynXf.reserve(nch * 12)); HIPCHK(e->d_dynAabb.reserve(nch * 6));
  DynMap dm = dynMapPoses(e);
  if (nch > 0) {
    hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, s, devScene(e), dm, e->d_objQT.p, 1, e->d, e->d_dynXf.p,
                       e->d_dynAabb.p);
    HIPCHK(hipGetLastError());
  }
  return launchRays(e, dm, from, to, n_rays, closest_fraction);
}

// The allocation-free variant: caller-owned DEVICE pointers in and out, working buffers in `scratch` (through the
// per-device pointer), three kernels on the engine's stream, no hipMalloc / hipFree / allocating library call.  It does
// not touch the engine's own pose tables or slab, so staged DOF vectors and pending enqueues are left as they are.
int tohc_ray_cast_poses_noalloc(tohc_engine *e, const double *d_obj_T, const double *d_from, const double *d_to, int n_rays,
                                float *d_closest_fraction) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene) return fail(e, TOHC_E_STATE, "upload the scene first");
  if (!d_obj_T || n_rays < 0 || (n_rays > 0 && (!d_from || !d_to || !d_closest_fraction))) return fail(e, TOHC_E_INVALID, "null pose, ray or result pointer");
  if (e->n_children > TOHC_NOALLOC_MAX_CHILDREN)
    return fail(e, TOHC_E_CAPACITY, "ray_cast_poses_noalloc: the scene has %d convex geometries, the fixed working buffer holds %d", e->n_children,
                TOHC_NOALLOC_MAX_CHILDREN);
  if (n_rays > TOHC_NOALLOC_MAX_RAYS)
    return fail(e, TOHC_E_CAPACITY, "ray_cast_poses_noalloc: %d rays in one call, the fixed working buffer holds %d", n_rays, TOHC_NOALLOC_MAX_RAYS);
  if (n_rays == 0) return TOHC_OK;
  HIPCHK(hipSetDevice(e->device));
  float *sp = (e->device >= 0 && e->device < SCRATCH_MAX_DEVICES) ? g_scratch_ptr[e->device] : nullptr;
  if (!sp) return fail(e, TOHC_E_STATE, "ray_cast_poses_noalloc: no working buffer on device %d", e->device);
  float *xf = sp, *box = sp + SCRATCH_XF_FLOATS, *rays = box + SCRATCH_BOX_FLOATS;
  hipStream_t s = e->stream;
  const size_t nch = e->n_children;
  const DynMap dm = dynMapPoses(e);
  // every word the kernels below read from the working buffer is written by the two kernels before them in this call:
  // nothing of an earlier call's contents is used, so nothing needs clearing
  if (nch > 0) hipLaunchKernelGGL(k_dyn_children, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, s, devScene(e), dm, d_obj_T, 1, e->d, xf, box);
  hipLaunchKernelGGL(k_ray_prep, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, s, d_from, d_to, n_rays, rays,
                     reinterpret_cast<unsigned int *>(d_closest_fraction));   // (presets every fraction to 1 = nothing hit)
  const size_t tot = (size_t)n_rays * nch;   // <= 65536 * 8192 tests: the grid below fits
  if (tot > 0) {
    RayArgs ra;
    ra.sc = devScene(e); ra.dm = dm; ra.dynXf = xf; ra.rays = rays; ra.closest = reinterpret_cast<unsigned int *>(d_closest_fraction); ra.n_rays = n_rays;
    hipLaunchKernelGGL(k_ray, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, ra);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  return TOHC_OK;
}

int tohc_ray_cast(tohc_engine *e, const double *dofs, const double *from, const double *to, int n_rays, float *closest_fraction) {
  if (!e) return TOHC_E_INVALID;
  if (!e->have_scene || !e->have_robot) return fail(e, TOHC_E_STATE, "upload the scene and the robot first");
  if (!dofs || n_rays < 0 || (n_rays > 0 && (!from || !to || !closest_fraction))) return fail(e, TOHC_E_INVALID, "null DOF, ray or result pointer");
  if (n_rays == 0) return TOHC_OK;
  int rc = tohc_sync(e);
  if (rc != TOHC_OK) return rc;
  rc = tohc_stage_dofs(e, dofs, 1, 1);
  if (rc != TOHC_OK) return rc;
  hipStream_t s = e->stream;
  DynMap dm = dynMapRobot(e);
  hipLaunchKernelGGL(k_fk, dim3(1), dim3(64), (size_t)e->n_links * 7 * sizeof(double), s, devRobot(e), devScene(e), dm, e->d_dofs.p, 1, e->d,
                     e->d_linkQT.p, e->d_objQT.p, e->d_dynXf.p, e->d_dynAabb.p);
  HIPCHK(hipGetLastError());
  e->fk_done = false;   // no enqueue may append to these one-set tables as if they were a trajectory's
  e->staged = false;
  return launchRays(e, dm, from, to, n_rays, closest_fraction);
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
  hipLaunchKernelGGL(k_fk, dim3((unsigned)n_cfg), dim3(64), (size_t)e->n_links * 7 * sizeof(double), s, devRobot(e), devScene(e), dynMapRobot(e),
                     e->d_dofs.p, n_cfg, e->d, e->d_linkQT.p, (double *)nullptr, (float *)nullptr, (float *)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(link_T, e->d_linkQT.p, (size_t)n_cfg * e->n_links * 7 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return TOHC_OK;
}

}  // extern "C"
