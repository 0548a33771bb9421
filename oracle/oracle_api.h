/*
 * oracle_api.h -- plain-C scene description and result records shared by the two CPU checkers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped engine: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these libraries, and only as the
 * checker.  The product path (trajopt_amd/, include/) never links or imports them.
 *
 *   oracle/trajopt_oracle.c   -> oracle/liboracle.so      CPU restatement (travels to the GPU box)
 *   oracle/ref_driver.cpp     -> oracle/_ref/libref.so    the reference's own vendored Bullet, compiled from
 *                                                         /root/reference in the build container only
 *
 * The scene format is "what OpenRAVE would hand to BulletCollisionChecker", flattened:
 *   shape   = one convex geometry (reference: createShapePrimitive, bullet_collision_checker.cpp:134-212)
 *   child   = one geometry of a link: a shape + its local transform (geom->GetTransform())
 *   object  = one link with geometry = one Bullet collision object (CollisionObjectFromLink, :215-255)
 * Transforms are OpenRAVE style: 7 doubles (qw qx qy qz tx ty tz); the checkers round them to fp32 exactly
 * as toBt() does (bullet_collision_checker.cpp:72-83).
 */
#ifndef TRAJOPT_ORACLE_API_H
#define TRAJOPT_ORACLE_API_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_BOX = 0, ORA_HULL = 1, ORA_SPHERE = 2, ORA_CYLZ = 3 };
enum { ORA_GROUP_ROBOT = 1, ORA_GROUP_KINBODY = 2 }; /* collision_checker.hpp:32-35 */
enum { ORA_CC_NONE = 0, ORA_CC_TIME0 = 1, ORA_CC_TIME1 = 2, ORA_CC_BETWEEN = 3 }; /* collision_checker.hpp:12-17 */

typedef struct ora_scene {
  int n_shapes;
  const int *shape_kind;     /* [n_shapes] ORA_*                                            */
  const float *shape_param;  /* [n_shapes*4] box: half extents xyz; sphere: r; cylZ: r, h/2 */
  const int *shape_voff;     /* [n_shapes] first vertex (hulls)                             */
  const int *shape_vcnt;     /* [n_shapes] vertex count (hulls, <= 64)                      */
  int n_verts;
  const float *verts;        /* [n_verts*3] xyz, fp32, exactly the points given to addPoint */
  int n_children;
  const int *child_shape;    /* [n_children]                                                */
  const double *child_T;     /* [n_children*7] local transform of the geometry in its link  */
  int n_objects;
  const int *obj_child_off;  /* [n_objects]                                                 */
  const int *obj_child_cnt;  /* [n_objects]                                                 */
  const int *obj_compound;   /* [n_objects] 0: bare convex shape (one geom at identity), 1: btCompoundShape */
  const int *obj_group;      /* [n_objects] ORA_GROUP_*                                     */
} ora_scene;

/* One contact, in the reference's Collision layout (collision_checker.hpp:19-29) with links replaced by
 * (object, child) ids.  A/B follow Bullet's GJK roles (cast contacts: after the collector's swap, B is
 * the swept robot link). */
typedef struct ora_contact {
  int objA, childA, objB, childB;
  int cctype;
  float time;
  double dist;
  double n[3];   /* normalB2A */
  double ptA[3], ptB[3], ptB1[3];
} ora_contact;

/* One hit of ContinuousCheckTrajectory (bullet_collision_checker.cpp:683-753): the reference builds
 * Collision(link, hitLink, hitPoint, hitPoint, hitNormal, 0, 1, seg + fraction) (whose ctor then drops the time). */
typedef struct ora_sweep_hit {
  int obj, child;       /* the swept link and its geometry                         */
  int seg;              /* trajectory segment seg -> seg+1                         */
  int hitObj, hitChild; /* closest object hit (the reference reports the link only) */
  float fraction;       /* m_closestHitFraction                                    */
  double n[3], pt[3];   /* m_hitNormalWorld, m_hitPointWorld                       */
} ora_sweep_hit;

/* Flattened kinematic model.  OpenRAVE (FK, Jacobians) is absent from /root/reference and from this image, so the
 * conventions below are the build's own, chosen to follow what configuration_space.cpp:12-23,83-133 relies on:
 *   link i hangs off link parent[i] (parents precede children; parent -1 = the base link);
 *   T_world_link = T_world_parent * T_origin[i] * Motion(jtype[i], axis[i], value)   (URDF convention)
 *   value = mult[i]*q_robot[dof[i]] + offs[i]; revolute: rotation about axis by value, prismatic: translation
 *   the optimisation vector is [q_robot[joint_inds[0..nj)], affine values in the order X, Y, Z, RotationAxis]
 *   (RaveGetTransformFromAffineDOFValues order); with affine DOFs the base transform starts from identity
 *   (configuration_space.cpp:14-16), else it is base_T.
 * Transforms compose in fp64 quaternion form (q = qa*qb renormalised, t = qa.rotate(tb)+ta). */
enum { ORA_J_FIXED = 0, ORA_J_REVOLUTE = 1, ORA_J_PRISMATIC = 2 };
enum { ORA_DOF_X = 1, ORA_DOF_Y = 2, ORA_DOF_Z = 4, ORA_DOF_ROTAXIS = 8 };
/* dof[i] of a FIXED link: marks the link of a body the robot has grabbed, hanging off its grabber link.
 * RobotAndDOF::GetAffectedLinks lists such links unconditionally, with the grabber's index as their Jacobian link
 * (configuration_space.cpp:120-131): they are queried even when no active DOF moves them. */
#define ORA_DOF_GRABBED (-2)
typedef struct ora_robot {
  int n_links;
  const int *parent;       /* [n_links]                                                       */
  const int *jtype;        /* [n_links] ORA_J_*                                               */
  const double *T_origin;  /* [n_links*7]                                                     */
  const double *axis;      /* [n_links*3] unit, in the link's own (joint) frame               */
  const int *dof;          /* [n_links] robot DOF driving the joint, -1 if fixed              */
  const double *mult;      /* [n_links]                                                       */
  const double *offs;      /* [n_links]                                                       */
  const int *link_obj;     /* [n_links] scene object holding the link's geometry, or -1       */
  int n_dof;
  const double *q_robot;   /* [n_dof] current values of all robot DOFs (inactive ones stay)   */
  double base_T[7];
  int n_joint_inds;
  const int *joint_inds;   /* [n_joint_inds] active robot DOFs                                 */
  int affine;              /* ORA_DOF_* mask                                                   */
  double rot_axis[3];
} ora_robot;

/* One linearised contact: the reference's Collision plus the AffExpr pieces of collision_terms.cpp:37-80.
 *   discrete:  expr(x)      = const0 + grad0 . x_t
 *   cast:      expr(x0, x1) = (1-time)*(const0 + grad0 . x_t) + time*(const1 + grad1 . x_t1)
 * grad0/grad1 live in a separate [n][2][K] array. */
typedef struct ora_lin {
  int t;           /* timestep (cast: the first of the pair) */
  ora_contact c;
  double const0, const1;
} ora_lin;

#ifdef __cplusplus
}
#endif
#endif
