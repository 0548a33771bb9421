/*
 * trajopt_hipcc.h -- C-ABI of libtrajopt_hipcc.so, the MI355X (gfx950) collision-cost linearization engine.
 *
 * This is the drop-in boundary for ONE path of joschu/trajopt: everything underneath
 *   trajopt::CollisionChecker::{LinksVsAll, CastVsAll}      src/trajopt/collision_checker.hpp:40-84
 *   CollisionsToDistanceExpressions                         src/trajopt/collision_terms.cpp:37-80
 * A C++ adapter `HIPCollisionChecker : trajopt::CollisionChecker` binds these entry points; no
 * OpenRAVE, Bullet or torch type crosses the boundary: plain pointers and sizes only.  Every function returns 0 on
 * success or a negative TOHC_E_* code; tohc_last_error() gives the message (no exceptions cross; the reference
 * throws std::runtime_error, collision_checker.hpp:57-62, which the adapter re-creates from the code).
 *
 * Threading: like the reference checker (one per environment, not re-entrant) a handle serialises its calls; use one
 * handle per host thread / per GPU.
 *
 * Transforms are OpenRAVE style, 7 doubles (qw qx qy qz tx ty tz); the engine rounds them to fp32 and builds the
 * rotation matrix exactly as toBt() + btMatrix3x3::setRotation do (bullet_collision_checker.cpp:72-83).
 */
#ifndef TRAJOPT_HIPCC_H
#define TRAJOPT_HIPCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tohc_engine tohc_engine; /* opaque */

enum {
  TOHC_OK = 0,
  TOHC_E_INVALID = -1,   /* bad argument / inconsistent sizes            */
  TOHC_E_HIP = -2,       /* a HIP runtime call failed                    */
  TOHC_E_CAPACITY = -3,  /* pair list or result slab too small           */
  TOHC_E_STATE = -4,     /* call order (e.g. query before upload)        */
  TOHC_E_UNSUPPORTED = -5
};

enum { TOHC_BOX = 0, TOHC_HULL = 1, TOHC_SPHERE = 2, TOHC_CYLZ = 3 };  /* createShapePrimitive, bullet_collision_checker.cpp:134-212; param[0]: box hx / sphere radius / cylinder radius */
enum { TOHC_GROUP_ROBOT = 1, TOHC_GROUP_KINBODY = 2 };  /* CollisionFilterGroups, collision_checker.hpp:32-35        */
enum { TOHC_CC_NONE = 0, TOHC_CC_TIME0 = 1, TOHC_CC_TIME1 = 2, TOHC_CC_BETWEEN = 3 }; /* CastCollisionType, :12-17  */
enum { TOHC_J_FIXED = 0, TOHC_J_REVOLUTE = 1, TOHC_J_PRISMATIC = 2 };
/* dof[i] of a TOHC_J_FIXED link: the link of a body the robot has GRABBED, hanging off its grabber link (T_origin = the
 * pose in the grabber's frame).  RobotAndDOF::GetAffectedLinks lists such links unconditionally, with the grabber's index
 * as their Jacobian link (configuration_space.cpp:120-131): they are query links even when no active DOF moves them. */
#define TOHC_DOF_GRABBED (-2)
enum { TOHC_DOF_X = 1, TOHC_DOF_Y = 2, TOHC_DOF_Z = 4, TOHC_DOF_ROTAXIS = 8 }; /* OpenRAVE DOFAffine, configuration_space.cpp:47-78 */

/* One linearised contact = the reference's Collision (collision_checker.hpp:19-29) + the AffExpr pieces of
 * CollisionsToDistanceExpressions (collision_terms.cpp:37-80).  152 bytes; gradients live in a parallel
 * double[n][2][K] array.
 *   discrete:  dist(x)      ~ const0 + grad0 . x_t
 *   cast:      dist(x0, x1) ~ (1-time)*(const0 + grad0 . x_t) + time*(const1 + grad1 . x_{t+gap})
 * idA/idB are object (link) ids, childA/childB geometry indices inside the link; A/B follow Bullet's GJK roles
 * (for cast contacts B is the swept robot link, bullet_collision_checker.cpp:913-917). */
typedef struct tohc_record {
  int32_t problem, t;
  int32_t idA, childA, idB, childB;
  int32_t cctype;
  float time;
  double dist;
  double n[3];               /* normalB2A */
  double ptA[3], ptB[3], ptB1[3];
  double const0, const1;
} tohc_record;

typedef struct tohc_counters {
  /* since the last sync (all enqueues it covered) */
  int64_t aabb_tests;   /* (timestep, query child, other child) tuples examined by the broadphase    */
  int64_t pair_evals;   /* tuples that survived the box cull: signed-distance pair evaluations         */
  int64_t epa_evals;    /* of those, how many went through GJK2+EPA                                   */
  int64_t contacts;     /* records in the slab                                                        */
  double ms_last_call;  /* device time of those enqueues (HIP events on the engine's stream)          */
  /* cumulative since tohc_create; per-kernel device time from HIP events around every launch         */
  int64_t cum_aabb_tests, cum_pair_evals, cum_epa_evals, cum_launches;
  double cum_ms_fk, cum_ms_children, cum_ms_broadphase, cum_ms_gjk, cum_ms_epa;
  /* pair evaluations the separating-axis pre-test inside the broadphase kernel could not finish ("farther than d")
   * and that therefore ran the iterative GJK; gjk_evals <= pair_evals */
  int64_t gjk_evals, cum_gjk_evals;
  double cum_ms_finish; /* k_finish (cast classification, records, gradient rows); cum_ms_epa is k_pen + k_epa */
  /* work done, for the flop side of the roofline: passes of the GJK loop body in k_gjk (each one
   * Minkowski support query = one hull scan per side), and support queries of the penetration solver (GJK2 and the
   * Distance fall-back in k_pen; EPA iterations and EncloseOrigin in k_epa).  Since the last sync / cumulative. */
  int64_t gjk_iters, pen_supports, epa_supports;
  int64_t cum_gjk_iters, cum_pen_supports, cum_epa_supports;
  /* EPA runs that left the small first-attempt pools of k_tail_epa and were repeated on the reference's full pools
   * (64 support vertices, 128 faces): same results either way, this only says how often the short cut missed */
  int64_t epa_redone, cum_epa_redone;
  /* penetration pairs whose class shard of the k_tail_pen -> k_tail_epa queue was full and that went to their block's
   * shard instead (the queue is filed by how long the pair's EPA run is expected to be; same results either way) */
  int64_t pen_refiled, cum_pen_refiled;
} tohc_counters;

/* replaces CreateCollisionChecker(env)                       src/trajopt/bullet_collision_checker.cpp:1036-1039 */
int tohc_create(tohc_engine **out, int device);
void tohc_destroy(tohc_engine *e);
const char *tohc_last_error(const tohc_engine *e);

/* replaces AddKinBody / CollisionObjectFromLink / createShapePrimitive   bullet_collision_checker.cpp:134-255,535-561
 *   shapes:   kind, param[4] (box: half extents as OpenRAVE gives them), vertex range into the SoA vertex arrays
 *   children: one geometry of a link = shape + local transform (geom->GetTransform())
 *   objects:  one link with geometry; compound!=0 <=> btCompoundShape (not "one geometry at identity");
 *             group = RobotFilter / KinBodyFilter; static_T = world transform used when the object is not driven
 *             by the uploaded robot. */
int tohc_upload_scene(tohc_engine *e, int n_shapes, const int32_t *shape_kind, const float *shape_param /*[n][4]*/,
                      const int32_t *shape_voff, const int32_t *shape_vcnt, int n_verts, const float *vx, const float *vy,
                      const float *vz, int n_children, const int32_t *child_shape, const double *child_T /*[n][7]*/,
                      int n_objects, const int32_t *obj_child_off, const int32_t *obj_child_cnt, const int32_t *obj_compound,
                      const int32_t *obj_group, const double *obj_static_T /*[n][7]*/);

/* replaces RobotAndDOF (configuration_space.hpp) + OpenRAVE's FK / CalculateActiveJacobian  configuration_space.cpp:12-23,83-133
 * Flattened joint tree, parents first; conventions in DESIGN.md ("FK model").  link_obj[i] = scene object carrying the
 * link's geometry (-1: none); grabbed bodies: see TOHC_DOF_GRABBED.  The optimisation vector is [q[joint_inds]..., affine X, Y, Z, RotationAxis]. */
int tohc_upload_robot(tohc_engine *e, int n_links, const int32_t *parent, const int32_t *jtype, const double *T_origin /*[n][7]*/,
                      const double *axis /*[n][3]*/, const int32_t *dof, const double *mult, const double *offs,
                      const int32_t *link_obj, int n_dof, const double *q_robot, const double *base_T /*[7]*/,
                      int n_joint_inds, const int32_t *joint_inds, int affine, const double *rot_axis /*[3]*/);

/* replaces m_allowedCollisionMatrix / Exclude|IncludeCollisionPair       bullet_collision_checker.cpp:342-351,606-627
 * allowed[i*n_objects+j] != 0 <=> object i (as query) may collide with object j. */
int tohc_set_allowed_matrix(tohc_engine *e, int n_objects, const uint8_t *allowed);

/* replaces BulletCollisionChecker::SetContactDistance / GetContactDistance    bullet_collision_checker.cpp:449-460 */
int tohc_set_contact_distance(tohc_engine *e, float d);
double tohc_get_contact_distance(const tohc_engine *e);

/* replaces LinksVsAll per timestep + SingleTimestepCollisionEvaluator::CalcDistExpressions
 *                     bullet_collision_checker.cpp:500-524, collision_terms.cpp:118-136
 * dofs: [B][T][K] fp64, B independent problems (multi-start) over the same scene and robot.  Query links are the
 * robot's affected links with geometry (GetAffectedLinks, configuration_space.cpp:110-118).  filter_mask as in
 * LinksVsAll (-1: everything, incl. self collision).  want_jacobians=0 skips the gradient epilogue (CalcDists).
 * On return *records / *grads point into a pinned result slab owned by the engine, valid until the next call on
 * this handle; records are in canonical order (problem, t, idA, childA, idB, childB). */
int tohc_links_vs_all_batch(tohc_engine *e, const double *dofs, int B, int T, int filter_mask, int want_jacobians,
                            const tohc_record **records, const double **grads /*[n][2][K]*/, int *n_records);

/* replaces CastVsAll per timestep pair + CastCollisionEvaluator::CalcDistExpressions
 *                     bullet_collision_checker.cpp:974-1023, collision_terms.cpp:64-80,156-168
 * one swept check per (problem, t, t+gap), t in [0, T-gap); record.t is the first timestep of the pair. */
int tohc_cast_vs_all_batch(tohc_engine *e, const double *dofs, int B, int T, int gap, int want_jacobians,
                           const tohc_record **records, const double **grads, int *n_records);

/* The batch calls above are stage -> enqueue -> fetch in a row.  Callers that keep inputs resident (an SQP loop that
 * re-linearizes, the benchmark) use the pieces: stage copies the DOF vectors to HBM; enqueue runs FK, broadphase and
 * narrowphase asynchronously on the engine's stream and leaves the slab on the device (append != 0 adds to the slab
 * of the previous enqueue, so one slab can carry the discrete and the cast cost of one SQP linearization); fetch
 * waits, copies the slab to the pinned host buffer and puts it in canonical order.  No counterpart in the reference
 * (its checker is synchronous, one timestep per call: SURVEY.md F6). */
int tohc_stage_dofs(tohc_engine *e, const double *dofs, int B, int T);
int tohc_enqueue_links_vs_all(tohc_engine *e, int filter_mask, int want_jacobians, int append);
int tohc_enqueue_cast_vs_all(tohc_engine *e, int gap, int want_jacobians, int append);

/* One SQP linearization of a problem with a discrete and a continuous collision term (CollisionCostInfo::hatch,
 * problem_description.cpp:516-529: one CollisionCost per timestep, over (i) or over (i, i + gap); the two constructors,
 * collision_terms.cpp:196-205, make a SingleTimestepCollisionEvaluator or a CastCollisionEvaluator; python_examples/
 * benchmark.py:76-118 asks for both kinds).  A NEW slab holding exactly what
 *   tohc_enqueue_links_vs_all(e, filter_mask, want_jacobians, 0);  tohc_enqueue_cast_vs_all(e, gap, want_jacobians, 1);
 * enqueue -- same records, gradient rows and counters -- with one broadphase launch for both passes where the engine's
 * set-up allows (the cast pass's query boxes hold the discrete pass's). */
int tohc_enqueue_linearization(tohc_engine *e, int filter_mask, int gap, int want_jacobians);
/* The same for ONE SHARD of a trajectory cut by timestep over several GPUs ("timesteps block-partitioned with
 * a 1-pose halo").  The reference makes one collision cost per timestep and one per (t, t + gap), each independent of the
 * others (problem_description.cpp:516-529; CastCollisionEvaluator reads exactly the variables of its two timesteps,
 * collision_terms.cpp:156-161), so a shard that owns timesteps [lo, hi) stages the poses [lo, min(hi + gap, T)) and asks
 * for the discrete pass over its own n_discrete = hi - lo of them and the cast pass over every staged pair: the last `gap`
 * poses are the halo, the second pose of its last cast pairs and some other shard's discrete timesteps.  1 <= n_discrete
 * <= the staged timestep count (equal: the shard that holds the trajectory's end, same as tohc_enqueue_linearization).
 * record.t counts from the shard's first pose; the union of the shards' slabs, t shifted by lo, is the single-GPU slab
 * (trajopt_amd/multi_gpu.py: timestep_shard, merge_timestep_shards). */
int tohc_enqueue_linearization_window(tohc_engine *e, int filter_mask, int gap, int want_jacobians, int n_discrete);
int tohc_sync(tohc_engine *e);
int tohc_fetch(tohc_engine *e, const tohc_record **records, const double **grads, int *n_records);

/* tohc_fetch without the copy to the host: the slab in canonical order as it lies in DEVICE memory (d_grads rows of
 * *row_doubles = 2K doubles, or null / 0 without gradient rows), valid until the next enqueue on this handle.  For the
 * north-star's one exchange step -- the RCCL all-gather of every GPU's linearizations -- which takes
 * device buffers: no round trip through host memory.  No counterpart in the reference (single process, no GPU). */
int tohc_fetch_device(tohc_engine *e, const tohc_record **d_records, const double **d_grads, int *n_records, int *row_doubles);

/* replaces, for the collision costs of one linearization, everything between the contacts and the solver's input:
 *   CollisionsToDistanceExpressions  src/trajopt/collision_terms.cpp:37-80     CollisionCost::convex  :205-214
 *   ConvexObjective::addHinge        src/sco/modeling.cpp:22-29                BPMPDModel::optimize   src/sco/bpmpd_interface.cpp:320-372
 * (SURVEY.md 8 f2).  The slab of the last enqueues (one discrete pass and / or one cast pass, with gradient rows) is put in
 * canonical order and turned, ON THE DEVICE, into the rows and columns BPMPD takes; the host reads those instead of the
 * 152 + 16 K bytes of every contact.  For problem p, with x[t*K + k] its trajectory variables and one new variable
 * h_i >= 0 per contact i (rows row_off[p] .. row_off[p+1], canonical order):
 *     row i:   dist_pen - (const_i + grad_i . x) - h_i <= 0          objective += coeff * h_i
 *   rhs[i]     = -(dist_pen - const_i)                                (bpmpd rhs = -aff.constant; row bounds [-BIG, 0])
 *   column (p, t, k) = entries col_off[c] .. col_off[c+1], c = (p*T + t)*K + k:  (col_idx = row within the problem,
 *     0-based and ascending; col_val = -(d dist_i / d x[t*K+k])), exact zeros dropped as sco's simplify2 does -- acolcnt,
 *     acolidx (before its +1), acolnzs of those variables.  Each h_i's own column is the single entry (i, -1).
 *   dist[i] is the signed distance (CollisionCost::value = sum coeff * max(0, dist_pen - dist)); row_t[i] = the timestep
 *     (cast: the first of the pair) | TOHC_HINGE_CAST: which cost object of the reference the row belongs to.
 * Values are bit-identical to the reference's fp64 arithmetic on the same contacts.  All pointers are pinned host
 * memory owned by the engine.
 *
 * tohc_fetch_hinges            = begin + end.
 * tohc_fetch_hinges_begin      waits for the slab, starts sort + assembly + the copy of the per-row arrays on a stream of
 *                              its own and returns; the next tohc_stage_dofs / enqueues may follow at once (they wait on
 *                              the device only until the assembly has read the slab).
 * tohc_fetch_hinges_end        waits for those, copies the columns (their size is known only now) and returns the block:
 *                              called after the next linearization has been enqueued, the copy runs while the GPU
 *                              computes.  One begin may be outstanding; a block stays valid until the second next begin
 *                              (two host buffer sets alternate). */
#define TOHC_HINGE_CAST (1 << 30)
typedef struct tohc_hinges {
  int32_t n_problems, T, K, gap;
  int64_t n_rows, nnz;
  const int64_t *row_off;   /* [n_problems + 1]          */
  const double *rhs;        /* [n_rows]                  */
  const double *dist;       /* [n_rows]                  */
  const int32_t *row_t;     /* [n_rows]                  */
  const int64_t *col_off;   /* [n_problems * T * K + 1]  */
  const int32_t *col_idx;   /* [nnz]                     */
  const double *col_val;    /* [nnz]                     */
} tohc_hinges;
int tohc_fetch_hinges(tohc_engine *e, double dist_pen_discrete, double dist_pen_cast, tohc_hinges *out);
int tohc_fetch_hinges_begin(tohc_engine *e, double dist_pen_discrete, double dist_pen_cast);
int tohc_fetch_hinges_end(tohc_engine *e, tohc_hinges *out);

/* replaces LinksVsAll / CastVsAll on explicit poses (no robot model needed): the reference reads poses from the
 * OpenRAVE environment at call time (UpdateBulletFromRave, bullet_collision_checker.cpp:648-653).
 * obj_T: [n_sets][n_objects][7]; query: object ids.  No Jacobians (no kinematic model). */
int tohc_links_vs_all_poses(tohc_engine *e, const double *obj_T, int n_sets, const int32_t *query, int n_query, int filter_mask,
                            const tohc_record **records, int *n_records);
/* cast_T1: [n_sets][n_query][7] = pose of each query link at the end of the sweep (start = its pose in obj_T). */
int tohc_cast_vs_all_poses(tohc_engine *e, const double *obj_T, const double *cast_T1, int n_sets, const int32_t *query,
                           int n_query, const tohc_record **records, int *n_records);

/* replaces BulletCollisionChecker::ContinuousCheckTrajectory                 bullet_collision_checker.cpp:683-753
 * (btCollisionWorld::convexSweepTest -> btContinuousConvexCollision::calcTimeOfImpact per link geometry and segment).
 * dofs: [B][T][K]; one record at most per (problem, segment t -> t+1, affected link geometry): the closest
 * KinBodyFilter object hit on the way.  record: idA/childA = the swept link geometry, idB/childB = what it hits,
 * time = hit fraction in [0,1] of the segment (the reference passes t + fraction to Collision's ctor), n = hit normal,
 * ptA = ptB = hit point, dist = 0; no gradients.  Like the reference it ignores the allowed-pair matrix. */
int tohc_continuous_check_trajectory(tohc_engine *e, const double *dofs, int B, int T, const tohc_record **records, int *n_records);
/* the same on explicit poses: obj_T [n_steps][n_objects][7]; the query links sweep from step s to s+1 against the
 * other objects' poses at step s. */
int tohc_continuous_check_poses(tohc_engine *e, const double *obj_T, int n_steps, const int32_t *query, int n_query,
                                const tohc_record **records, int *n_records);

/* replaces BulletCollisionChecker::RayCastCollision                           bullet_collision_checker.cpp:352-360
 * (btCollisionWorld::rayTest with a ClosestRayResultCallback -> btSubsimplexConvexCast::calcTimeOfImpact per convex
 * geometry), for a batch of rays against ONE configuration.  from / to: [n_rays][3] world points;
 * closest_fraction[i] = the callback's m_closestHitFraction in [0, 1]: 1 = nothing hit, so the reference's return value
 * (hasHit()) is closest_fraction[i] < 1.  Every object takes part (robot links too); like the reference it ignores the
 * allowed-pair matrix and the contact distance.  dofs: [K], the robot's configuration (the reference reads it from the
 * OpenRAVE environment at call time, UpdateBulletFromRave :356). */
int tohc_ray_cast(tohc_engine *e, const double *dofs, const double *from, const double *to, int n_rays, float *closest_fraction);
/* the same on explicit poses: obj_T [n_objects][7] */
int tohc_ray_cast_poses(tohc_engine *e, const double *obj_T, const double *from, const double *to, int n_rays, float *closest_fraction);

/* NO COUNTERPART IN THE REFERENCE.  tohc_ray_cast_poses for callers that cannot allocate (tight serving loops, harnesses
 * that forbid hipMalloc, code that must not disturb a HIP graph capture going on elsewhere in the process -- the call
 * itself runs on the engine's private stream and waits for it, so it is not something to capture): the same kernels and the same answers, but every
 * pointer is DEVICE memory the caller owns (d_obj_T [n_objects][7], d_from / d_to [n_rays][3] doubles,
 * d_closest_fraction [n_rays] floats), and the working buffers (child poses, cull boxes, the fp32 ray table) live in a
 * fixed array inside the library's device image instead of the engine's grow-only buffers: the call makes no
 * hipMalloc / hipFree / allocating library call.  The results are complete when it returns (it waits for its stream).
 * Limits: the scene may hold at most TOHC_NOALLOC_MAX_CHILDREN convex geometries and a call at most
 * TOHC_NOALLOC_MAX_RAYS rays, else TOHC_E_CAPACITY and nothing runs.  Every engine on one GPU shares that GPU's
 * array (about 2.7 MB, held until the process ends): calls that overlap in time on one GPU race -- serialise them. */
#define TOHC_NOALLOC_MAX_CHILDREN 8192
#define TOHC_NOALLOC_MAX_RAYS 65536
int tohc_ray_cast_poses_noalloc(tohc_engine *e, const double *d_obj_T, const double *d_from, const double *d_to, int n_rays,
                                float *d_closest_fraction);

/* device FK only (diagnostics / tests): link_T [B*T][n_links][7] */
int tohc_forward_kinematics(tohc_engine *e, const double *dofs, int n_cfg, double *link_T);

int tohc_get_counters(const tohc_engine *e, tohc_counters *out);
/* The per-stage times of tohc_counters -- cum_ms_fk / _broadphase / _gjk / _epa / _finish -- come from HIP events recorded
 * between the kernels of every enqueue; each is a marker packet the next kernel of the stream waits behind.  on = 0 leaves
 * them out (those fields then stay as they are; ms_last_call, the span of the call, is still measured): a single
 * problem's linearization answers ~35 us sooner, a 64-problem step ~2 %.  Default on; TOHC_TIMING=0 in the environment
 * sets the default off.  No counterpart in the reference. */
int tohc_set_kernel_timing(tohc_engine *e, int on);

/* capacity knobs (defaults sized for SURVEY config 4 on one MI355X).  max_pairs bounds, per enqueue, the pairs handed to
 * the iterative solver AND the pre-test's intermediate list (pairs whose bounding-sphere test could not separate them,
 * about twice as many); max_records the contact records of one slab.  Overflow = TOHC_E_CAPACITY, nothing is truncated. */
int tohc_set_capacity(tohc_engine *e, int64_t max_pairs, int64_t max_records);

/* replaces BulletCollisionChecker::AllVsAll                   src/trajopt/bullet_collision_checker.cpp:463-498, 415-422
 * (performDiscreteCollisionDetection + a walk over the dispatcher's persistent manifolds) at the given object poses
 * [n_objects][7].  Every unordered pair of objects once -- the allowed matrix is read at (lower, higher) object number,
 * as the near callback's CanCollide(proxy0, proxy1), and again at (A, B) as the manifold walk does (:484) -- with
 * Bullet's roles for A and B; contacts are kept up to the
 * manifolds' breaking threshold 2.001 d (there is no CollisionCollector on this path), and dist / ptA / ptB have been
 * through the manifold's local-frame round trip (btManifoldResult::addContactPoint, refreshContactPoints), as the
 * reference reads them.  Every convex pair owns a manifold with one point, so the reference's weight 1 / numContacts is
 * 1 for every record.  (The reference's manifolds also remember the contact distance they were created under; this is
 * the result of a checker whose distance was set before the first call.)  Records as tohc_links_vs_all_poses, no
 * gradient rows. */
int tohc_all_vs_all_poses(tohc_engine *e, const double *obj_T, const tohc_record **records, int *n_records);

/* replaces btShapeHull(&btConvexTriangleMeshShape).buildHull() as createShapePrimitive calls it for a triangle mesh
 *                                                            src/trajopt/bullet_collision_checker.cpp:174-203
 *          ext/bullet/src/BulletCollision/CollisionShapes/btShapeHull.cpp:38-100, LinearMath/btConvexHull.cpp:691-812
 * The engine takes the FINAL points of a TOHC_HULL shape; for a mesh this is how to get them.  verts: n_verts x 3,
 * tri: n_tri x 3 vertex numbers (the mesh's index list, in its order: ties between support vertices go to the corner
 * named first).  Meshes of fewer than 50 vertices are used as they are, larger ones are reduced to the hull of their 42
 * support points (<= 42 points, values and order as the reference's btConvexHullShape holds them; fp32 host code, no
 * GPU, no engine).  Writes min(result, cap) points to out and returns the result's point count (call with cap = 0 to
 * ask), or TOHC_E_INVALID. */
int tohc_shape_hull(const float *verts, int n_verts, const int32_t *tri, int n_tri, float *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
