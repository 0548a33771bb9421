/*
 * trajopt_oracle.c -- CPU restatement of the collision-cost linearization path of joschu/trajopt.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_api.h).  Plain C99, fp32 for everything Bullet computes in btScalar,
 * fp64 for FK / Jacobians / linearization, scalar, single threaded, compiled with -ffp-contract=off so every
 * multiply and add rounds separately like the reference build.  Pinned against (a) the reference's unit-test
 * known answers and (b) the reference's own vendored Bullet compiled into oracle/_ref/libref.so
 * (tests/test_oracle_vs_ref.py, tests/golden/).  FK / Jacobian parity is UNPINNED against OpenRAVE (absent
 * from /root/reference and from this image); it is pinned by finite differences and closed-form cases instead.
 *
 * Reference files followed (paths relative to /root/reference):
 *   [bcc]  src/trajopt/bullet_collision_checker.cpp
 *   [ct]   src/trajopt/collision_terms.cpp
 *   [cs]   src/trajopt/configuration_space.cpp
 *   [gjk]  ext/bullet/src/BulletCollision/NarrowPhaseCollision/btGjkPairDetector.cpp
 *   [vss]  ext/bullet/src/BulletCollision/NarrowPhaseCollision/btVoronoiSimplexSolver.{h,cpp}
 *   [epa]  ext/bullet/src/BulletCollision/NarrowPhaseCollision/btGjkEpa2.cpp
 *   [pds]  ext/bullet/src/BulletCollision/NarrowPhaseCollision/btGjkEpaPenetrationDepthSolver.cpp
 *   [cvx]  ext/bullet/src/BulletCollision/CollisionShapes/btConvexShape.cpp
 *   [cca]  ext/bullet/src/BulletCollision/CollisionDispatch/btConvexConvexAlgorithm.cpp
 *   [cpa]  ext/bullet/src/BulletCollision/CollisionDispatch/btCompoundCollisionAlgorithm.cpp
 *   [cw]   ext/bullet/src/BulletCollision/CollisionDispatch/btCollisionWorld.cpp
 *   [lm]   ext/bullet/src/LinearMath/{btVector3.h,btMatrix3x3.h,btTransform.h,btScalar.h,btAabbUtil2.h}
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_api.h"

/* ------------------------------------------------------------------------------------------------------ */
/* fp32 vector algebra in btVector3 / btMatrix3x3 operation order [lm]                                      */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { v3 r[3]; } m3;           /* rows */
typedef struct { m3 b; v3 o; } xf;        /* btTransform */

#define SIMD_EPSILON FLT_EPSILON
#define BT_LARGE_FLOAT 1e18f

static v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* btVector3.h:229-247 */
static float vlen2(v3 a) { return vdot(a, a); }
static float vlen(v3 a) { return sqrtf(vlen2(a)); }
static v3 vdiv(v3 a, float s) { return vscale(a, 1.0f / s); }                /* btVector3.h:838 */
static v3 vcross(v3 a, v3 b) {                                                /* btVector3.h:380-384 */
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static int veq(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static v3 mulMV(const m3 *m, v3 v) { return V(vdot(m->r[0], v), vdot(m->r[1], v), vdot(m->r[2], v)); }
static float tdotx(const m3 *m, v3 v) { return m->r[0].x * v.x + m->r[1].x * v.y + m->r[2].x * v.z; }
static float tdoty(const m3 *m, v3 v) { return m->r[0].y * v.x + m->r[1].y * v.y + m->r[2].y * v.z; }
static float tdotz(const m3 *m, v3 v) { return m->r[0].z * v.x + m->r[1].z * v.y + m->r[2].z * v.z; }
static v3 mulVM(v3 v, const m3 *m) { return V(tdotx(m, v), tdoty(m, v), tdotz(m, v)); } /* btMatrix3x3.h:1187 */
static m3 mulMM(const m3 *a, const m3 *b) {                                   /* btMatrix3x3.h:1270 */
  m3 r;
  for (int i = 0; i < 3; ++i) r.r[i] = V(tdotx(b, a->r[i]), tdoty(b, a->r[i]), tdotz(b, a->r[i]));
  return r;
}
static m3 transposeTimes(const m3 *a, const m3 *m) {                          /* btMatrix3x3.h:1074-1084 */
  m3 r;
  const v3 *e = a->r, *n = m->r;
  r.r[0] = V(e[0].x * n[0].x + e[1].x * n[1].x + e[2].x * n[2].x, e[0].x * n[0].y + e[1].x * n[1].y + e[2].x * n[2].y,
             e[0].x * n[0].z + e[1].x * n[1].z + e[2].x * n[2].z);
  r.r[1] = V(e[0].y * n[0].x + e[1].y * n[1].x + e[2].y * n[2].x, e[0].y * n[0].y + e[1].y * n[1].y + e[2].y * n[2].y,
             e[0].y * n[0].z + e[1].y * n[1].z + e[2].y * n[2].z);
  r.r[2] = V(e[0].z * n[0].x + e[1].z * n[1].x + e[2].z * n[2].x, e[0].z * n[0].y + e[1].z * n[1].y + e[2].z * n[2].y,
             e[0].z * n[0].z + e[1].z * n[1].z + e[2].z * n[2].z);
  return r;
}
static v3 xfApply(const xf *t, v3 x) { return vadd(mulMV(&t->b, x), t->o); }  /* btTransform.h:94-97 */
static xf xfMul(const xf *a, const xf *b) {                                   /* btTransform.h:236-242 */
  xf r;
  r.b = mulMM(&a->b, &b->b);
  r.o = xfApply(a, b->o);
  return r;
}
static xf xfInverseTimes(const xf *a, const xf *t) {                          /* btTransform.h:228-234 */
  xf r;
  v3 v = vsub(t->o, a->o);
  r.b = transposeTimes(&a->b, &t->b);
  r.o = mulVM(v, &a->b);
  return r;
}
/* toBt(OR::Transform) [bcc:72-83] + btMatrix3x3::setRotation scalar branch [btMatrix3x3.h:204-269] */
static xf xfFrom7(const double *t) {
  float qx = (float)t[1], qy = (float)t[2], qz = (float)t[3], qw = (float)t[0];
  float d = qx * qx + qy * qy + qz * qz + qw * qw; /* btQuaternion::length2: x,y,z,w order */
  float s = 2.0f / d;
  float xs = qx * s, ys = qy * s, zs = qz * s;
  float wx = qw * xs, wy = qw * ys, wz = qw * zs;
  float xx = qx * xs, xy = qx * ys, xz = qx * zs;
  float yy = qy * ys, yz = qy * zs, zz = qz * zs;
  xf r;
  r.b.r[0] = V(1.0f - (yy + zz), xy - wz, xz + wy);
  r.b.r[1] = V(xy + wz, 1.0f - (xx + zz), yz - wx);
  r.b.r[2] = V(xz - wy, yz + wx, 1.0f - (xx + yy));
  r.o = V((float)t[4], (float)t[5], (float)t[6]);
  return r;
}

/* ------------------------------------------------------------------------------------------------------ */
/* shapes and support functions                                                                             */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct {
  int kind;
  v3 dims;           /* box: m_implicitShapeDimensions after the margin round trip; cylZ: (r, r, h/2) likewise */
  float margin;      /* getMargin(): 0 for box/hull/cylinder (MARGIN, bcc:23), radius for spheres           */
  int nv;
  const float *v;    /* hull vertices xyz */
  v3 lmin, lmax;     /* exact local AABB, used by the engine-defined cull (not Bullet's cached one)          */
} shape_t;

/* A convex operand of GJK: a plain shape, or the swept hull of one (CastHullShape, bcc:816-882). */
typedef struct {
  const shape_t *s;
  int cast;
  xf t01;
} cvx_t;

static float fsel(float a, float b, float c) { return a >= 0 ? b : c; } /* btScalar.h:468-473 */

static v3 hullMaxDot(const shape_t *s, v3 d) { /* btVector3::maxDot scalar branch, btVector3.h:1003-1018 */
  float best = -INFINITY;
  int idx = -1;
  for (int i = 0; i < s->nv; ++i) {
    float dot = s->v[3 * i] * d.x + s->v[3 * i + 1] * d.y + s->v[3 * i + 2] * d.z;
    if (dot > best) { best = dot; idx = i; }
  }
  if (idx < 0) idx = 0; /* NaN direction: Bullet would index -1; unreachable for finite input */
  return V(s->v[3 * idx], s->v[3 * idx + 1], s->v[3 * idx + 2]);
}

static v3 cylZSupport(v3 he, v3 v) { /* [cvx:156-215] with upAxis 2: XX=0, YY=2, ZZ=1 */
  float radius = he.x, halfHeight = he.z;
  float s = sqrtf(v.x * v.x + v.y * v.y);
  if (s != 0.0f) {
    float d = radius / s;
    return V(v.x * d, v.y * d, (double)v.z < 0.0 ? -halfHeight : halfHeight);
  }
  return V(radius, 0.0f, (double)v.z < 0.0 ? -halfHeight : halfHeight);
}

/* btConvexShape::localGetSupportVertexWithoutMarginNonVirtual for the plain kinds [cvx:120-296] */
static v3 supPlainNV(const shape_t *s, v3 d) {
  switch (s->kind) {
    case ORA_SPHERE: return V(0, 0, 0);
    case ORA_BOX:
      return V(fsel(-d.x, -s->dims.x, s->dims.x), fsel(-d.y, -s->dims.y, s->dims.y), fsel(-d.z, -s->dims.z, s->dims.z));
    case ORA_CYLZ: return cylZSupport(s->dims, d);
    default: return hullMaxDot(s, d); /* localScaling == (1,1,1): the multiply is exact */
  }
}

static v3 normalizedOrMinus1(v3 d) { /* [cvx:298-306], btSphereShape.cpp:43-48 */
  if (vlen2(d) < (SIMD_EPSILON * SIMD_EPSILON)) d = V(-1, -1, -1);
  return vdiv(d, vlen(d));
}

/* the VIRTUAL localGetSupportingVertex (with margin) that CastHullShape calls on its inner shape */
static v3 supPlainVirtual(const shape_t *s, v3 d) {
  switch (s->kind) {
    case ORA_BOX: /* btBoxShape.h:50-59: tie rule d >= 0 -> +h (differs from the non-virtual one) */
      return V(fsel(d.x, s->dims.x + s->margin, -(s->dims.x + s->margin)), fsel(d.y, s->dims.y + s->margin, -(s->dims.y + s->margin)),
               fsel(d.z, s->dims.z + s->margin, -(s->dims.z + s->margin)));
    case ORA_SPHERE: return vscale(normalizedOrMinus1(d), s->margin);
    case ORA_CYLZ: {
      v3 sv = cylZSupport(s->dims, d);
      if (s->margin != 0.0f) sv = vadd(sv, vscale(normalizedOrMinus1(d), s->margin));
      return sv;
    }
    default: {
      v3 sv = hullMaxDot(s, d);
      if (s->margin != 0.0f) sv = vadd(sv, vscale(normalizedOrMinus1(d), s->margin));
      return sv;
    }
  }
}

/* localGetSupportVertexWithoutMarginNonVirtual for a GJK operand */
static v3 supNV(const cvx_t *c, v3 d) {
  if (!c->cast) return supPlainNV(c->s, d);
  /* CastHullShape::localGetSupportingVertex [bcc:826-830] */
  v3 sv0 = supPlainVirtual(c->s, d);
  v3 sv1 = xfApply(&c->t01, supPlainVirtual(c->s, mulVM(d, &c->t01.b)));
  return (vdot(d, sv0) > vdot(d, sv1)) ? sv0 : sv1;
}
static float cvxMargin(const cvx_t *c) { return c->cast ? 0.0f : c->s->margin; }
/* localGetSupportVertexNonVirtual [cvx:298-308] (EPA "with margins" path) */
static v3 supWithMarginNV(const cvx_t *c, v3 d) {
  v3 n = normalizedOrMinus1(d);
  return vadd(supNV(c, n), vscale(n, cvxMargin(c)));
}

/* ------------------------------------------------------------------------------------------------------ */
/* btVoronoiSimplexSolver [vss]                                                                              */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct {
  v3 closest;
  int used[4];
  float bc[4];
  int degenerate;
} subres_t;

typedef struct {
  int n;
  v3 W[5], P[5], Q[5];
  v3 cP1, cP2, cV, lastW;
  float eqThresh;
  int cachedValid;
  subres_t bc;
  int needsUpdate;
} vss_t;

static void subresReset(subres_t *r) {
  r->degenerate = 0;
  r->bc[0] = r->bc[1] = r->bc[2] = r->bc[3] = 0;
  r->used[0] = r->used[1] = r->used[2] = r->used[3] = 0;
}
static void setBC(subres_t *r, float a, float b, float c, float d) { r->bc[0] = a; r->bc[1] = b; r->bc[2] = c; r->bc[3] = d; }
static int subresValid(const subres_t *r) { return r->bc[0] >= 0 && r->bc[1] >= 0 && r->bc[2] >= 0 && r->bc[3] >= 0; }

static void vssReset(vss_t *s) { /* [vss.cpp:66-73] */
  s->cachedValid = 0;
  s->n = 0;
  s->needsUpdate = 1;
  s->lastW = V(BT_LARGE_FLOAT, BT_LARGE_FLOAT, BT_LARGE_FLOAT);
  subresReset(&s->bc);
  s->eqThresh = 0.0001f; /* VORONOI_DEFAULT_EQUAL_VERTEX_THRESHOLD, vss.h:28-29,136 */
}
static void vssRemove(vss_t *s, int i) {
  s->n--;
  s->W[i] = s->W[s->n];
  s->P[i] = s->P[s->n];
  s->Q[i] = s->Q[s->n];
}
static void vssReduce(vss_t *s, const int *used) { /* [vss.cpp:45-59] */
  if (s->n >= 4 && !used[3]) vssRemove(s, 3);
  if (s->n >= 3 && !used[2]) vssRemove(s, 2);
  if (s->n >= 2 && !used[1]) vssRemove(s, 1);
  if (s->n >= 1 && !used[0]) vssRemove(s, 0);
}
static void vssAdd(vss_t *s, v3 w, v3 p, v3 q) {
  s->lastW = w;
  s->needsUpdate = 1;
  s->W[s->n] = w;
  s->P[s->n] = p;
  s->Q[s->n] = q;
  s->n++;
}
static int vssInSimplex(const vss_t *s, v3 w) { /* [vss.cpp:283-305], BT_USE_EQUAL_VERTEX_THRESHOLD is defined */
  int found = 0;
  for (int i = 0; i < s->n; ++i)
    if (vlen2(vsub(w, s->W[i])) <= s->eqThresh) found = 1;
  if (veq(w, s->lastW)) return 1;
  return found;
}

static void closestPtTriangle(v3 p, v3 a, v3 b, v3 c, subres_t *r) { /* [vss.cpp:330-423] */
  r->used[0] = r->used[1] = r->used[2] = r->used[3] = 0;
  v3 ab = vsub(b, a), ac = vsub(c, a), ap = vsub(p, a);
  float d1 = vdot(ab, ap), d2 = vdot(ac, ap);
  if (d1 <= 0.0f && d2 <= 0.0f) {
    r->closest = a; r->used[0] = 1; setBC(r, 1, 0, 0, 0);
    return;
  }
  v3 bp = vsub(p, b);
  float d3 = vdot(ab, bp), d4 = vdot(ac, bp);
  if (d3 >= 0.0f && d4 <= d3) {
    r->closest = b; r->used[1] = 1; setBC(r, 0, 1, 0, 0);
    return;
  }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    r->closest = vadd(a, vscale(ab, v));
    r->used[0] = r->used[1] = 1;
    setBC(r, 1 - v, v, 0, 0);
    return;
  }
  v3 cp = vsub(p, c);
  float d5 = vdot(ab, cp), d6 = vdot(ac, cp);
  if (d6 >= 0.0f && d5 <= d6) {
    r->closest = c; r->used[2] = 1; setBC(r, 0, 0, 1, 0);
    return;
  }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    r->closest = vadd(a, vscale(ac, w));
    r->used[0] = r->used[2] = 1;
    setBC(r, 1 - w, 0, w, 0);
    return;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    r->closest = vadd(b, vscale(vsub(c, b), w));
    r->used[1] = r->used[2] = 1;
    setBC(r, 0, 1 - w, w, 0);
    return;
  }
  float denom = 1.0f / (va + vb + vc);
  float v = vb * denom, w = vc * denom;
  r->closest = vadd(vadd(a, vscale(ab, v)), vscale(ac, w));
  r->used[0] = r->used[1] = r->used[2] = 1;
  setBC(r, 1 - v - w, v, w, 0);
}

static int pointOutsideOfPlane(v3 p, v3 a, v3 b, v3 c, v3 d) { /* [vss.cpp:430-454] */
  v3 normal = vcross(vsub(b, a), vsub(c, a));
  float signp = vdot(vsub(p, a), normal);
  float signd = vdot(vsub(d, a), normal);
  if (signd * signd < (1e-4f * 1e-4f)) return -1;
  return signp * signd < 0.0f;
}

static int closestPtTetrahedron(v3 p, v3 a, v3 b, v3 c, v3 d, subres_t *fr) { /* [vss.cpp:457-608] */
  subres_t tr;
  fr->closest = p;
  fr->used[0] = fr->used[1] = fr->used[2] = fr->used[3] = 1;
  int oABC = pointOutsideOfPlane(p, a, b, c, d);
  int oACD = pointOutsideOfPlane(p, a, c, d, b);
  int oADB = pointOutsideOfPlane(p, a, d, b, c);
  int oBDC = pointOutsideOfPlane(p, b, d, c, a);
  if (oABC < 0 || oACD < 0 || oADB < 0 || oBDC < 0) { fr->degenerate = 1; return 0; }
  if (!oABC && !oACD && !oADB && !oBDC) return 0;
  float best = FLT_MAX;
  if (oABC) {
    closestPtTriangle(p, a, b, c, &tr);
    v3 q = tr.closest;
    float sq = vdot(vsub(q, p), vsub(q, p));
    if (sq < best) {
      best = sq; fr->closest = q;
      fr->used[0] = tr.used[0]; fr->used[1] = tr.used[1]; fr->used[2] = tr.used[2]; fr->used[3] = 0;
      setBC(fr, tr.bc[0], tr.bc[1], tr.bc[2], 0);
    }
  }
  if (oACD) {
    closestPtTriangle(p, a, c, d, &tr);
    v3 q = tr.closest;
    float sq = vdot(vsub(q, p), vsub(q, p));
    if (sq < best) {
      best = sq; fr->closest = q;
      fr->used[0] = tr.used[0]; fr->used[1] = 0; fr->used[2] = tr.used[1]; fr->used[3] = tr.used[2];
      setBC(fr, tr.bc[0], 0, tr.bc[1], tr.bc[2]);
    }
  }
  if (oADB) {
    closestPtTriangle(p, a, d, b, &tr);
    v3 q = tr.closest;
    float sq = vdot(vsub(q, p), vsub(q, p));
    if (sq < best) {
      best = sq; fr->closest = q;
      fr->used[0] = tr.used[0]; fr->used[1] = tr.used[2]; fr->used[2] = 0; fr->used[3] = tr.used[1];
      setBC(fr, tr.bc[0], tr.bc[2], 0, tr.bc[1]);
    }
  }
  if (oBDC) {
    closestPtTriangle(p, b, d, c, &tr);
    v3 q = tr.closest;
    float sq = vdot(vsub(q, p), vsub(q, p));
    if (sq < best) {
      best = sq; fr->closest = q;
      fr->used[0] = 0; fr->used[1] = tr.used[0]; fr->used[2] = tr.used[2]; fr->used[3] = tr.used[1];
      setBC(fr, 0, tr.bc[0], tr.bc[2], tr.bc[1]);
    }
  }
  return 1;
}

static int vssUpdate(vss_t *s) { /* updateClosestVectorAndPoints [vss.cpp:90-240] */
  if (!s->needsUpdate) return s->cachedValid;
  subresReset(&s->bc);
  s->needsUpdate = 0;
  switch (s->n) {
    case 0: s->cachedValid = 0; break;
    case 1:
      s->cP1 = s->P[0];
      s->cP2 = s->Q[0];
      s->cV = vsub(s->cP1, s->cP2);
      subresReset(&s->bc);
      setBC(&s->bc, 1, 0, 0, 0);
      s->cachedValid = subresValid(&s->bc);
      break;
    case 2: {
      v3 from = s->W[0], to = s->W[1];
      v3 diff = vsub(V(0, 0, 0), from);
      v3 v = vsub(to, from);
      float t = vdot(v, diff);
      if (t > 0) {
        float dotVV = vdot(v, v);
        if (t < dotVV) {
          t /= dotVV;
          diff = vsub(diff, vscale(v, t));
          s->bc.used[0] = 1; s->bc.used[1] = 1;
        } else {
          t = 1;
          diff = vsub(diff, v);
          s->bc.used[1] = 1;
        }
      } else {
        t = 0;
        s->bc.used[0] = 1;
      }
      setBC(&s->bc, 1 - t, t, 0, 0);
      s->cP1 = vadd(s->P[0], vscale(vsub(s->P[1], s->P[0]), t));
      s->cP2 = vadd(s->Q[0], vscale(vsub(s->Q[1], s->Q[0]), t));
      s->cV = vsub(s->cP1, s->cP2);
      vssReduce(s, s->bc.used);
      s->cachedValid = subresValid(&s->bc);
      break;
    }
    case 3: {
      closestPtTriangle(V(0, 0, 0), s->W[0], s->W[1], s->W[2], &s->bc);
      const float *b = s->bc.bc;
      s->cP1 = vadd(vadd(vscale(s->P[0], b[0]), vscale(s->P[1], b[1])), vscale(s->P[2], b[2]));
      s->cP2 = vadd(vadd(vscale(s->Q[0], b[0]), vscale(s->Q[1], b[1])), vscale(s->Q[2], b[2]));
      s->cV = vsub(s->cP1, s->cP2);
      vssReduce(s, s->bc.used);
      s->cachedValid = subresValid(&s->bc);
      break;
    }
    case 4: {
      int sep = closestPtTetrahedron(V(0, 0, 0), s->W[0], s->W[1], s->W[2], s->W[3], &s->bc);
      if (sep) {
        const float *b = s->bc.bc;
        s->cP1 = vadd(vadd(vadd(vscale(s->P[0], b[0]), vscale(s->P[1], b[1])), vscale(s->P[2], b[2])), vscale(s->P[3], b[3]));
        s->cP2 = vadd(vadd(vadd(vscale(s->Q[0], b[0]), vscale(s->Q[1], b[1])), vscale(s->Q[2], b[2])), vscale(s->Q[3], b[3]));
        s->cV = vsub(s->cP1, s->cP2);
        vssReduce(s, s->bc.used);
      } else {
        if (s->bc.degenerate) {
          s->cachedValid = 0;
        } else {
          s->cachedValid = 1;
          s->cV = V(0, 0, 0);
        }
        break;
      }
      s->cachedValid = subresValid(&s->bc);
      break;
    }
    default: s->cachedValid = 0;
  }
  return s->cachedValid;
}

/* ------------------------------------------------------------------------------------------------------ */
/* btGjkEpaSolver2: GJK2 + EPA [epa]                                                                         */
/* ------------------------------------------------------------------------------------------------------ */
#define GJK_MAX_ITERATIONS 128
#define GJK_ACCURARY 0.0001f
#define GJK_MIN_DISTANCE 0.0001f
#define GJK_DUPLICATED_EPS 0.0001f
#define EPA_MAX_VERTICES 64
#define EPA_MAX_FACES (EPA_MAX_VERTICES * 2)
#define EPA_MAX_ITERATIONS 255
#define EPA_ACCURACY 0.0001f
#define EPA_PLANE_EPS 0.00001f

typedef struct {
  const cvx_t *sh[2];
  m3 toshape1;
  xf toshape0;
  int margins;
} mink_t;

static v3 minkSupport0(const mink_t *m, v3 d) { return m->margins ? supWithMarginNV(m->sh[0], d) : supNV(m->sh[0], d); }
static v3 minkSupport1(const mink_t *m, v3 d) {
  v3 dl = mulMV(&m->toshape1, d);
  v3 s = m->margins ? supWithMarginNV(m->sh[1], dl) : supNV(m->sh[1], dl);
  return xfApply(&m->toshape0, s);
}
static v3 minkSupport(const mink_t *m, v3 d) { return vsub(minkSupport0(m, d), minkSupport1(m, vneg(d))); }

typedef struct { v3 d, w; } sv_t;
typedef struct { sv_t *c[4]; float p[4]; unsigned rank; } simplex_t;
enum { G_VALID, G_INSIDE, G_FAILED };

typedef struct {
  mink_t shape;
  v3 ray;
  float distance;
  simplex_t simplices[2];
  sv_t store[4];
  sv_t *free_[4];
  unsigned nfree, current;
  simplex_t *simplex;
  int status;
} gjk2_t;

static void g2support(const gjk2_t *g, v3 d, sv_t *sv) {
  sv->d = vdiv(d, vlen(d));
  sv->w = minkSupport(&g->shape, sv->d);
}
static void g2remove(gjk2_t *g, simplex_t *s) { g->free_[g->nfree++] = s->c[--s->rank]; }
static void g2append(gjk2_t *g, simplex_t *s, v3 v) {
  s->p[s->rank] = 0;
  s->c[s->rank] = g->free_[--g->nfree];
  g2support(g, v, s->c[s->rank++]);
}
static float det3(v3 a, v3 b, v3 c) { /* [epa:387-392] */
  return (a.y * b.z * c.x + a.z * b.x * c.y - a.x * b.z * c.y - a.y * b.x * c.z + a.x * b.y * c.z - a.z * b.y * c.x);
}
static float projOrigin2(v3 a, v3 b, float *w, unsigned *m) { /* [epa:393-407] */
  v3 d = vsub(b, a);
  float l = vlen2(d);
  if (l > 0.0f) {
    float t = (l > 0 ? -vdot(a, d) / l : 0);
    if (t >= 1) { w[0] = 0; w[1] = 1; *m = 2; return vlen2(b); }
    else if (t <= 0) { w[0] = 1; w[1] = 0; *m = 1; return vlen2(a); }
    else { w[0] = 1 - (w[1] = t); *m = 3; return vlen2(vadd(a, vscale(d, t))); }
  }
  return -1;
}
static float projOrigin3(v3 a, v3 b, v3 c, float *w, unsigned *m) { /* [epa:408-453] */
  static const unsigned imd3[] = {1, 2, 0};
  const v3 vt[3] = {a, b, c};
  const v3 dl[3] = {vsub(a, b), vsub(b, c), vsub(c, a)};
  v3 n = vcross(dl[0], dl[1]);
  float l = vlen2(n);
  if (l > 0.0f) {
    float mindist = -1;
    float subw[2] = {0.f, 0.f};
    unsigned subm = 0;
    for (unsigned i = 0; i < 3; ++i) {
      if (vdot(vt[i], vcross(dl[i], n)) > 0) {
        unsigned j = imd3[i];
        float subd = projOrigin2(vt[i], vt[j], subw, &subm);
        if ((mindist < 0) || (subd < mindist)) {
          mindist = subd;
          *m = ((subm & 1) ? 1u << i : 0) + ((subm & 2) ? 1u << j : 0);
          w[i] = subw[0];
          w[j] = subw[1];
          w[imd3[j]] = 0;
        }
      }
    }
    if (mindist < 0) {
      float d = vdot(a, n);
      float s = sqrtf(l);
      v3 p = vscale(n, d / l);
      mindist = vlen2(p);
      *m = 7;
      w[0] = vlen(vcross(dl[1], vsub(b, p))) / s;
      w[1] = vlen(vcross(dl[2], vsub(c, p))) / s;
      w[2] = 1 - (w[0] + w[1]);
    }
    return mindist;
  }
  return -1;
}
static float projOrigin4(v3 a, v3 b, v3 c, v3 d, float *w, unsigned *m) { /* [epa:454-502] */
  static const unsigned imd3[] = {1, 2, 0};
  const v3 vt[4] = {a, b, c, d};
  const v3 dl[3] = {vsub(a, d), vsub(b, d), vsub(c, d)};
  float vl = det3(dl[0], dl[1], dl[2]);
  int ng = (vl * vdot(a, vcross(vsub(b, c), vsub(a, b)))) <= 0;
  if (ng && (fabsf(vl) > 0.0f)) {
    float mindist = -1;
    float subw[3] = {0.f, 0.f, 0.f};
    unsigned subm = 0;
    for (unsigned i = 0; i < 3; ++i) {
      unsigned j = imd3[i];
      float s = vl * vdot(d, vcross(dl[i], dl[j]));
      if (s > 0) {
        float subd = projOrigin3(vt[i], vt[j], d, subw, &subm);
        if ((mindist < 0) || (subd < mindist)) {
          mindist = subd;
          *m = ((subm & 1) ? 1u << i : 0) + ((subm & 2) ? 1u << j : 0) + ((subm & 4) ? 8 : 0);
          w[i] = subw[0];
          w[j] = subw[1];
          w[imd3[j]] = 0;
          w[3] = subw[2];
        }
      }
    }
    if (mindist < 0) {
      mindist = 0;
      *m = 15;
      w[0] = det3(c, b, d) / vl;
      w[1] = det3(a, c, d) / vl;
      w[2] = det3(b, a, d) / vl;
      w[3] = 1 - (w[0] + w[1] + w[2]);
    }
    return mindist;
  }
  return -1;
}

static int g2Evaluate(gjk2_t *g, const mink_t *shape, v3 guess) { /* [epa:183-305] */
  unsigned iterations = 0;
  float sqdist = 0, alpha = 0;
  v3 lastw[4];
  unsigned clastw = 0;
  g->free_[0] = &g->store[0]; g->free_[1] = &g->store[1]; g->free_[2] = &g->store[2]; g->free_[3] = &g->store[3];
  g->nfree = 4;
  g->current = 0;
  g->status = G_VALID;
  g->shape = *shape;
  g->distance = 0;
  g->simplices[0].rank = 0;
  g->ray = guess;
  float sqrl = vlen2(g->ray);
  g2append(g, &g->simplices[0], sqrl > 0 ? vneg(g->ray) : V(1, 0, 0));
  g->simplices[0].p[0] = 1;
  g->ray = g->simplices[0].c[0]->w;
  sqdist = sqrl;
  lastw[0] = lastw[1] = lastw[2] = lastw[3] = g->ray;
  do {
    unsigned next = 1 - g->current;
    simplex_t *cs = &g->simplices[g->current];
    simplex_t *ns = &g->simplices[next];
    float rl = vlen(g->ray);
    if (rl < GJK_MIN_DISTANCE) { g->status = G_INSIDE; break; }
    g2append(g, cs, vneg(g->ray));
    v3 w = cs->c[cs->rank - 1]->w;
    int found = 0;
    for (unsigned i = 0; i < 4; ++i)
      if (vlen2(vsub(w, lastw[i])) < GJK_DUPLICATED_EPS) { found = 1; break; }
    if (found) { g2remove(g, &g->simplices[g->current]); break; }
    else lastw[clastw = (clastw + 1) & 3] = w;
    float omega = vdot(g->ray, w) / rl;
    alpha = omega > alpha ? omega : alpha;
    if (((rl - alpha) - (GJK_ACCURARY * rl)) <= 0) { g2remove(g, &g->simplices[g->current]); break; }
    float weights[4];
    unsigned mask = 0;
    switch (cs->rank) {
      case 2: sqdist = projOrigin2(cs->c[0]->w, cs->c[1]->w, weights, &mask); break;
      case 3: sqdist = projOrigin3(cs->c[0]->w, cs->c[1]->w, cs->c[2]->w, weights, &mask); break;
      case 4: sqdist = projOrigin4(cs->c[0]->w, cs->c[1]->w, cs->c[2]->w, cs->c[3]->w, weights, &mask); break;
    }
    if (sqdist >= 0) {
      ns->rank = 0;
      g->ray = V(0, 0, 0);
      g->current = next;
      for (unsigned i = 0, ni = cs->rank; i < ni; ++i) {
        if (mask & (1u << i)) {
          ns->c[ns->rank] = cs->c[i];
          ns->p[ns->rank++] = weights[i];
          g->ray = vadd(g->ray, vscale(cs->c[i]->w, weights[i]));
        } else {
          g->free_[g->nfree++] = cs->c[i];
        }
      }
      if (mask == 15) g->status = G_INSIDE;
    } else {
      g2remove(g, &g->simplices[g->current]);
      break;
    }
    g->status = ((++iterations) < GJK_MAX_ITERATIONS) ? g->status : G_FAILED;
  } while (g->status == G_VALID);
  g->simplex = &g->simplices[g->current];
  switch (g->status) {
    case G_VALID: g->distance = vlen(g->ray); break;
    case G_INSIDE: g->distance = 0; break;
    default: break;
  }
  return g->status;
}

static int g2EncloseOrigin(gjk2_t *g) { /* [epa:306-370] */
  simplex_t *s = g->simplex;
  switch (s->rank) {
    case 1:
      for (unsigned i = 0; i < 3; ++i) {
        v3 axis = V(i == 0, i == 1, i == 2);
        g2append(g, s, axis);
        if (g2EncloseOrigin(g)) return 1;
        g2remove(g, s);
        g2append(g, s, vneg(axis));
        if (g2EncloseOrigin(g)) return 1;
        g2remove(g, s);
      }
      break;
    case 2: {
      v3 d = vsub(s->c[1]->w, s->c[0]->w);
      for (unsigned i = 0; i < 3; ++i) {
        v3 axis = V(i == 0, i == 1, i == 2);
        v3 p = vcross(d, axis);
        if (vlen2(p) > 0) {
          g2append(g, s, p);
          if (g2EncloseOrigin(g)) return 1;
          g2remove(g, s);
          g2append(g, s, vneg(p));
          if (g2EncloseOrigin(g)) return 1;
          g2remove(g, s);
        }
      }
      break;
    }
    case 3: {
      v3 n = vcross(vsub(s->c[1]->w, s->c[0]->w), vsub(s->c[2]->w, s->c[0]->w));
      if (vlen2(n) > 0) {
        g2append(g, s, n);
        if (g2EncloseOrigin(g)) return 1;
        g2remove(g, s);
        g2append(g, s, vneg(n));
        if (g2EncloseOrigin(g)) return 1;
        g2remove(g, s);
      }
      break;
    }
    case 4:
      if (fabsf(det3(vsub(s->c[0]->w, s->c[3]->w), vsub(s->c[1]->w, s->c[3]->w), vsub(s->c[2]->w, s->c[3]->w))) > 0) return 1;
      break;
  }
  return 0;
}

typedef struct face_s {
  v3 n;
  float d;
  sv_t *c[3];
  struct face_s *f[3];
  struct face_s *l[2];
  unsigned char e[3];
  unsigned char pass;
} face_t;
typedef struct { face_t *root; unsigned count; } flist_t;
typedef struct { face_t *cf, *ff; unsigned nf; } horizon_t;
enum { E_VALID, E_TOUCHING, E_DEGENERATED, E_NONCONVEX, E_INVALIDHULL, E_OUTOFFACES, E_OUTOFVERTICES, E_ACCURACY, E_FALLBACK, E_FAILED };

typedef struct {
  int status;
  simplex_t result;
  v3 normal;
  float depth;
  sv_t sv_store[EPA_MAX_VERTICES];
  face_t fc_store[EPA_MAX_FACES];
  unsigned nextsv;
  flist_t hull, stock;
} epa_t;

static void ebind(face_t *fa, unsigned ea, face_t *fb, unsigned eb) {
  fa->e[ea] = (unsigned char)eb; fa->f[ea] = fb;
  fb->e[eb] = (unsigned char)ea; fb->f[eb] = fa;
}
static void eappend(flist_t *l, face_t *f) {
  f->l[0] = 0;
  f->l[1] = l->root;
  if (l->root) l->root->l[0] = f;
  l->root = f;
  ++l->count;
}
static void eremove(flist_t *l, face_t *f) {
  if (f->l[1]) f->l[1]->l[0] = f->l[0];
  if (f->l[0]) f->l[0]->l[1] = f->l[1];
  if (f == l->root) l->root = f->l[1];
  --l->count;
}
static void epaInit(epa_t *e) { /* [epa:583-593] */
  e->status = E_FAILED;
  e->normal = V(0, 0, 0);
  e->depth = 0;
  e->nextsv = 0;
  e->hull.root = 0; e->hull.count = 0;
  e->stock.root = 0; e->stock.count = 0;
  for (unsigned i = 0; i < EPA_MAX_FACES; ++i) eappend(&e->stock, &e->fc_store[EPA_MAX_FACES - i - 1]);
}
static int getedgedist(face_t *face, sv_t *a, sv_t *b, float *dist) { /* [epa:698-733] */
  v3 ba = vsub(b->w, a->w);
  v3 n_ab = vcross(ba, face->n);
  float a_dot_nab = vdot(a->w, n_ab);
  if (a_dot_nab < 0) {
    float ba_l2 = vlen2(ba);
    float a_dot_ba = vdot(a->w, ba);
    float b_dot_ba = vdot(b->w, ba);
    if (a_dot_ba > 0) {
      *dist = vlen(a->w);
    } else if (b_dot_ba < 0) {
      *dist = vlen(b->w);
    } else {
      float a_dot_b = vdot(a->w, b->w);
      float t = (vlen2(a->w) * vlen2(b->w) - a_dot_b * a_dot_b) / ba_l2;
      *dist = sqrtf(t > 0.0f ? t : 0.0f);
    }
    return 1;
  }
  return 0;
}
static face_t *newface(epa_t *e, sv_t *a, sv_t *b, sv_t *c, int forced) { /* [epa:734-778] */
  if (e->stock.root) {
    face_t *face = e->stock.root;
    eremove(&e->stock, face);
    eappend(&e->hull, face);
    face->pass = 0;
    face->c[0] = a; face->c[1] = b; face->c[2] = c;
    face->n = vcross(vsub(b->w, a->w), vsub(c->w, a->w));
    float l = vlen(face->n);
    int v = l > EPA_ACCURACY;
    if (v) {
      if (!(getedgedist(face, a, b, &face->d) || getedgedist(face, b, c, &face->d) || getedgedist(face, c, a, &face->d)))
        face->d = vdot(a->w, face->n) / l;
      face->n = vdiv(face->n, l);
      if (forced || (face->d >= -EPA_PLANE_EPS)) return face;
      else e->status = E_NONCONVEX;
    } else
      e->status = E_DEGENERATED;
    eremove(&e->hull, face);
    eappend(&e->stock, face);
    return 0;
  }
  e->status = e->stock.root ? E_OUTOFVERTICES : E_OUTOFFACES;
  return 0;
}
static face_t *findbest(epa_t *e) {
  face_t *minf = e->hull.root;
  float mind = minf->d * minf->d;
  for (face_t *f = minf->l[1]; f; f = f->l[1]) {
    float sqd = f->d * f->d;
    if (sqd < mind) { minf = f; mind = sqd; }
  }
  return minf;
}
static int expand(epa_t *e, unsigned pass, sv_t *w, face_t *f, unsigned ei, horizon_t *h) { /* [epa:794-827] */
  static const unsigned i1m3[] = {1, 2, 0};
  static const unsigned i2m3[] = {2, 0, 1};
  if (f->pass != pass) {
    unsigned e1 = i1m3[ei];
    if ((vdot(f->n, w->w) - f->d) < -EPA_PLANE_EPS) {
      face_t *nf = newface(e, f->c[e1], f->c[ei], w, 0);
      if (nf) {
        ebind(nf, 0, f, ei);
        if (h->cf) ebind(h->cf, 1, nf, 2); else h->ff = nf;
        h->cf = nf;
        ++h->nf;
        return 1;
      }
    } else {
      unsigned e2 = i2m3[ei];
      f->pass = (unsigned char)pass;
      if (expand(e, pass, w, f->f[e1], f->e[e1], h) && expand(e, pass, w, f->f[e2], f->e[e2], h)) {
        eremove(&e->hull, f);
        eappend(&e->stock, f);
        return 1;
      }
    }
  }
  return 0;
}
static int epaEvaluate(epa_t *e, gjk2_t *g, v3 guess) { /* [epa:594-697] */
  simplex_t *simplex = g->simplex;
  if ((simplex->rank > 1) && g2EncloseOrigin(g)) {
    while (e->hull.root) {
      face_t *f = e->hull.root;
      eremove(&e->hull, f);
      eappend(&e->stock, f);
    }
    e->status = E_VALID;
    e->nextsv = 0;
    if (det3(vsub(simplex->c[0]->w, simplex->c[3]->w), vsub(simplex->c[1]->w, simplex->c[3]->w),
             vsub(simplex->c[2]->w, simplex->c[3]->w)) < 0) {
      sv_t *t = simplex->c[0]; simplex->c[0] = simplex->c[1]; simplex->c[1] = t;
      float tp = simplex->p[0]; simplex->p[0] = simplex->p[1]; simplex->p[1] = tp;
    }
    face_t *tetra[4];
    tetra[0] = newface(e, simplex->c[0], simplex->c[1], simplex->c[2], 1);
    tetra[1] = newface(e, simplex->c[1], simplex->c[0], simplex->c[3], 1);
    tetra[2] = newface(e, simplex->c[2], simplex->c[1], simplex->c[3], 1);
    tetra[3] = newface(e, simplex->c[0], simplex->c[2], simplex->c[3], 1);
    if (e->hull.count == 4) {
      face_t *best = findbest(e);
      face_t outer = *best;
      unsigned pass = 0, iterations = 0;
      ebind(tetra[0], 0, tetra[1], 0);
      ebind(tetra[0], 1, tetra[2], 0);
      ebind(tetra[0], 2, tetra[3], 0);
      ebind(tetra[1], 1, tetra[3], 2);
      ebind(tetra[1], 2, tetra[2], 1);
      ebind(tetra[2], 2, tetra[3], 1);
      e->status = E_VALID;
      for (; iterations < EPA_MAX_ITERATIONS; ++iterations) {
        if (e->nextsv < EPA_MAX_VERTICES) {
          horizon_t hz = {0, 0, 0};
          sv_t *w = &e->sv_store[e->nextsv++];
          int valid = 1;
          best->pass = (unsigned char)(++pass);
          g2support(g, best->n, w);
          float wdist = vdot(best->n, w->w) - best->d;
          if (wdist > EPA_ACCURACY) {
            for (unsigned j = 0; (j < 3) && valid; ++j) valid &= expand(e, pass, w, best->f[j], best->e[j], &hz);
            if (valid && (hz.nf >= 3)) {
              ebind(hz.cf, 1, hz.ff, 2);
              eremove(&e->hull, best);
              eappend(&e->stock, best);
              best = findbest(e);
              outer = *best;
            } else { e->status = E_INVALIDHULL; break; }
          } else { e->status = E_ACCURACY; break; }
        } else { e->status = E_OUTOFVERTICES; break; }
      }
      v3 projection = vscale(outer.n, outer.d);
      e->normal = outer.n;
      e->depth = outer.d;
      e->result.rank = 3;
      e->result.c[0] = outer.c[0]; e->result.c[1] = outer.c[1]; e->result.c[2] = outer.c[2];
      e->result.p[0] = vlen(vcross(vsub(outer.c[1]->w, projection), vsub(outer.c[2]->w, projection)));
      e->result.p[1] = vlen(vcross(vsub(outer.c[2]->w, projection), vsub(outer.c[0]->w, projection)));
      e->result.p[2] = vlen(vcross(vsub(outer.c[0]->w, projection), vsub(outer.c[1]->w, projection)));
      float sum = e->result.p[0] + e->result.p[1] + e->result.p[2];
      e->result.p[0] /= sum; e->result.p[1] /= sum; e->result.p[2] /= sum;
      return e->status;
    }
  }
  e->status = E_FALLBACK;
  e->normal = vneg(guess);
  float nl = vlen(e->normal);
  if (nl > 0) e->normal = vdiv(e->normal, nl);
  else e->normal = V(1, 0, 0);
  e->depth = 0;
  e->result.rank = 1;
  e->result.c[0] = simplex->c[0];
  e->result.p[0] = 1;
  return e->status;
}

typedef struct { v3 w0, w1, normal; float distance; } epares_t;

static void minkInit(mink_t *m, const cvx_t *s0, const xf *w0, const cvx_t *s1, const xf *w1, int margins) { /* [epa:832-848] */
  m->sh[0] = s0; m->sh[1] = s1;
  m->toshape1 = transposeTimes(&w1->b, &w0->b);
  m->toshape0 = xfInverseTimes(w0, w1);
  m->margins = margins;
}

static long g_epa_iters, g_epa_calls;

static int solverPenetration(const cvx_t *s0, const xf *w0, const cvx_t *s1, const xf *w1, v3 guess, epares_t *r) { /* [epa:903-945] */
  mink_t shape;
  minkInit(&shape, s0, w0, s1, w1, 1);
  r->w0 = r->w1 = V(0, 0, 0);
  gjk2_t g;
  int st = g2Evaluate(&g, &shape, vneg(guess));
  if (st == G_INSIDE) {
    static epa_t e; /* ~12 KB; single threaded */
    epaInit(&e);
    g_epa_calls++;
    int es = epaEvaluate(&e, &g, vneg(guess));
    if (es != E_FAILED) {
      v3 w0v = V(0, 0, 0);
      for (unsigned i = 0; i < e.result.rank; ++i) w0v = vadd(w0v, vscale(minkSupport0(&shape, e.result.c[i]->d), e.result.p[i]));
      r->w0 = xfApply(w0, w0v);
      r->w1 = xfApply(w0, vsub(w0v, vscale(e.normal, e.depth)));
      r->normal = vneg(e.normal);
      r->distance = -e.depth;
      return 1;
    }
  }
  return 0;
}
static int solverDistance(const cvx_t *s0, const xf *w0, const cvx_t *s1, const xf *w1, v3 guess, epares_t *r) { /* [epa:865-900] */
  mink_t shape;
  minkInit(&shape, s0, w0, s1, w1, 0);
  r->w0 = r->w1 = V(0, 0, 0);
  gjk2_t g;
  int st = g2Evaluate(&g, &shape, guess);
  if (st == G_VALID) {
    v3 a = V(0, 0, 0), b = V(0, 0, 0);
    for (unsigned i = 0; i < g.simplex->rank; ++i) {
      float p = g.simplex->p[i];
      a = vadd(a, vscale(minkSupport0(&shape, g.simplex->c[i]->d), p));
      b = vadd(b, vscale(minkSupport1(&shape, vneg(g.simplex->c[i]->d)), p));
    }
    r->w0 = xfApply(w0, a);
    r->w1 = xfApply(w0, b);
    r->normal = vsub(a, b);
    r->distance = vlen(r->normal);
    r->normal = vdiv(r->normal, r->distance > GJK_MIN_DISTANCE ? r->distance : 1);
    return 1;
  }
  return 0;
}
/* btGjkEpaPenetrationDepthSolver::calcPenDepth [pds:24-64] */
static int calcPenDepth(const cvx_t *A, const cvx_t *B, const xf *tA, const xf *tB, v3 *v, v3 *wA, v3 *wB) {
  v3 guess = vsub(tA->o, tB->o);
  epares_t r;
  if (solverPenetration(A, tA, B, tB, guess, &r)) {
    *wA = r.w0; *wB = r.w1; *v = r.normal;
    return 1;
  }
  if (solverDistance(A, tA, B, tB, guess, &r)) {
    *wA = r.w0; *wB = r.w1; *v = r.normal;
    return 0;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------ */
/* btGjkPairDetector::getClosestPointsNonVirtual [gjk:82-452]                                                */
/* ------------------------------------------------------------------------------------------------------ */
#define REL_ERROR2 1.0e-6f
typedef struct { int valid; v3 normalOnB, pointOnB; float distance; int iters, degenerate, method; } gjkout_t;

static long g_gjk_calls, g_gjk_iters;

static void gjkPair(const cvx_t *A, const cvx_t *B, const xf *tA, const xf *tB, float maxDist2, gjkout_t *out) {
  float distance = 0;
  v3 normalInB = V(0, 0, 0), pointOnA = V(0, 0, 0), pointOnB = V(0, 0, 0);
  xf lA = *tA, lB = *tB;
  v3 offset = vscale(vadd(lA.o, lB.o), 0.5f);
  lA.o = vsub(lA.o, offset);
  lB.o = vsub(lB.o, offset);
  float marginA = cvxMargin(A), marginB = cvxMargin(B);
  int curIter = 0;
  const int maxIter = 1000;
  v3 sep = V(0, 1, 0);
  int isValid = 0, checkSimplex = 0, degenerate = 0, method = -1;
  float squaredDistance = BT_LARGE_FLOAT, delta = 0;
  float margin = marginA + marginB;
  vss_t sx;
  memset(&sx, 0, sizeof sx);
  vssReset(&sx);
  g_gjk_calls++;
  for (;;) {
    v3 sepA = mulVM(vneg(sep), &tA->b);
    v3 sepB = mulVM(sep, &tB->b);
    v3 pInA = supNV(A, sepA);
    v3 qInB = supNV(B, sepB);
    v3 pW = xfApply(&lA, pInA);
    v3 qW = xfApply(&lB, qInB);
    v3 w = vsub(pW, qW);
    delta = vdot(sep, w);
    if ((delta > 0.0f) && (delta * delta > squaredDistance * maxDist2)) { degenerate = 10; checkSimplex = 1; break; }
    if (vssInSimplex(&sx, w)) { degenerate = 1; checkSimplex = 1; break; }
    float f0 = squaredDistance - delta;
    float f1 = squaredDistance * REL_ERROR2;
    if (f0 <= f1) {
      degenerate = (f0 <= 0.0f) ? 2 : 11;
      checkSimplex = 1;
      break;
    }
    vssAdd(&sx, w, pW, qW);
    v3 newSep;
    int ok = vssUpdate(&sx);
    newSep = sx.cV;
    if (!ok) { degenerate = 3; checkSimplex = 1; break; }
    if (vlen2(newSep) < REL_ERROR2) { sep = newSep; degenerate = 6; checkSimplex = 1; break; }
    float prev = squaredDistance;
    squaredDistance = vlen2(newSep);
    if (prev - squaredDistance <= SIMD_EPSILON * prev) { checkSimplex = 1; degenerate = 12; break; }
    sep = newSep;
    if (curIter++ > maxIter) break;
    if (sx.n == 4) { degenerate = 13; break; }
  }
  g_gjk_iters += curIter;
  if (checkSimplex) {
    vssUpdate(&sx);
    pointOnA = sx.cP1;
    pointOnB = sx.cP2;
    normalInB = sep;
    float lenSqr = vlen2(sep);
    if ((double)lenSqr < 0.0001) degenerate = 5;
    if (lenSqr > SIMD_EPSILON * SIMD_EPSILON) {
      float rlen = 1.0f / sqrtf(lenSqr);
      normalInB = vscale(normalInB, rlen);
      float s = sqrtf(squaredDistance);
      pointOnA = vsub(pointOnA, vscale(sep, marginA / s));
      pointOnB = vadd(pointOnB, vscale(sep, marginB / s));
      distance = (1.0f / rlen) - margin;
      isValid = 1;
      method = 1;
    } else
      method = 2;
  }
  int catchDeg = degenerate && ((double)(distance + margin) < 0.01);
  if (!isValid || catchDeg) {
    v3 tmpA = V(0, 0, 0), tmpB = V(0, 0, 0);
    sep = V(0, 0, 0);
    int isValid2 = calcPenDepth(A, B, &lA, &lB, &sep, &tmpA, &tmpB);
    if (isValid2) {
      v3 tmpN = vsub(tmpB, tmpA);
      float lenSqr = vlen2(tmpN);
      if (lenSqr <= (SIMD_EPSILON * SIMD_EPSILON)) { tmpN = sep; lenSqr = vlen2(sep); }
      if (lenSqr > (SIMD_EPSILON * SIMD_EPSILON)) {
        tmpN = vdiv(tmpN, sqrtf(lenSqr));
        float distance2 = -vlen(vsub(tmpA, tmpB));
        if (!isValid || (distance2 < distance)) {
          distance = distance2; pointOnA = tmpA; pointOnB = tmpB; normalInB = tmpN; isValid = 1; method = 3;
        } else
          method = 8;
      } else
        method = 9;
    } else {
      if (vlen2(sep) > 0.0f) {
        float distance2 = vlen(vsub(tmpA, tmpB)) - margin;
        if (!isValid || (distance2 < distance)) {
          distance = distance2;
          pointOnA = tmpA;
          pointOnB = tmpB;
          pointOnA = vsub(pointOnA, vscale(sep, marginA));
          pointOnB = vadd(pointOnB, vscale(sep, marginB));
          normalInB = sep;
          normalInB = vdiv(normalInB, vlen(normalInB));
          isValid = 1;
          method = 6;
        } else
          method = 5;
      }
    }
  }
  out->valid = 0;
  out->iters = curIter;
  out->degenerate = degenerate;
  out->method = method;
  (void)pointOnA;
  if (isValid && ((distance < 0) || (distance * distance < maxDist2))) {
    out->valid = 1;
    out->normalOnB = normalInB;
    out->pointOnB = vadd(pointOnB, offset);
    out->distance = distance;
  }
}

/* ------------------------------------------------------------------------------------------------------ */
/* world glue                                                                                                */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct {
  int n_shapes, n_children, n_objects;
  shape_t *shapes;
  int *child_shape;
  xf *child_T;
  int *obj_off, *obj_cnt, *obj_compound, *obj_group;
  unsigned char *allowed;
  float *verts;     /* owned copy of the hull vertices */
  float d;          /* contact distance as float (SetContactDistance takes float) */
  float breaking;   /* gContactBreakingThreshold = 2.001*d (double product rounded to float) */
  int all_pairs;    /* inside ora_all_vs_all: contacts are kept up to the breaking threshold */
  long pair_evals, aabb_tests;
} world_t;

static float safeMargin(float cur, float x, float y, float z) {
  float mn = x < y ? (x < z ? x : z) : (y < z ? y : z); /* btVector3::minAxis */
  float safe = 0.1f * mn;
  return safe < cur ? safe : cur;
}

static void shapeInit(shape_t *s, const ora_scene *sc, int i) {
  memset(s, 0, sizeof *s);
  s->kind = sc->shape_kind[i];
  const float *p = sc->shape_param + 4 * i;
  const float CDM = 0.04f; /* CONVEX_DISTANCE_MARGIN: default m_collisionMargin at construction */
  switch (s->kind) {
    case ORA_BOX: {
      /* btBoxShape ctor: setSafeMargin (margin = min(0.04, 0.1*min half extent), btConvexInternalShape.h:70-85), then
       * dims = h*scaling - margin (btBoxShape.cpp:17-26); the checker's setMargin(0) gives (dims + margin) - 0
       * (btBoxShape.h:87-96) */
      float m = safeMargin(CDM, p[0], p[1], p[2]);
      s->dims = V(((p[0] * 1.0f - m) + m) - 0.0f, ((p[1] * 1.0f - m) + m) - 0.0f, ((p[2] * 1.0f - m) + m) - 0.0f);
      s->margin = 0;
      s->lmin = vneg(s->dims); s->lmax = s->dims;
      break;
    }
    case ORA_CYLZ: {
      float r = p[0], h = p[1];
      float m = safeMargin(CDM, r, r, h); /* btCylinderShape.cpp:18-27, same scheme */
      s->dims = V(((r * 1.0f - m) + m) - 0.0f, ((r * 1.0f - m) + m) - 0.0f, ((h * 1.0f - m) + m) - 0.0f);
      s->margin = 0;
      s->lmin = vneg(s->dims); s->lmax = s->dims;
      break;
    }
    case ORA_SPHERE:
      /* btSphereShape: implicit dims.x = margin = radius; setMargin(MARGIN=0) is applied by the checker
       * [bcc:227,244] but btSphereShape::setMargin -> btConvexInternalShape::setMargin sets m_collisionMargin = 0
       * while getMargin() returns getRadius() = m_implicitShapeDimensions.x * m_localScaling.x */
      s->dims = V(p[0], p[0], p[0]);
      s->margin = p[0] * 1.0f;
      s->lmin = V(-p[0], -p[0], -p[0]); s->lmax = V(p[0], p[0], p[0]);
      break;
    default: {
      s->nv = sc->shape_vcnt[i];
      s->v = sc->verts + 3 * sc->shape_voff[i];
      s->margin = 0;
      s->lmin = V(INFINITY, INFINITY, INFINITY);
      s->lmax = V(-INFINITY, -INFINITY, -INFINITY);
      for (int k = 0; k < s->nv; ++k) {
        const float *q = s->v + 3 * k;
        s->lmin = V(fminf(s->lmin.x, q[0]), fminf(s->lmin.y, q[1]), fminf(s->lmin.z, q[2]));
        s->lmax = V(fmaxf(s->lmax.x, q[0]), fmaxf(s->lmax.y, q[1]), fmaxf(s->lmax.z, q[2]));
      }
    }
  }
}

void *ora_create(const ora_scene *sc) {
  world_t *w = (world_t *)calloc(1, sizeof *w);
  w->n_shapes = sc->n_shapes; w->n_children = sc->n_children; w->n_objects = sc->n_objects;
  w->shapes = (shape_t *)calloc(sc->n_shapes ? sc->n_shapes : 1, sizeof(shape_t));
  /* vertices are copied so the world owns its data */
  float *verts = (float *)malloc(sizeof(float) * 3 * (sc->n_verts ? sc->n_verts : 1));
  if (sc->n_verts) memcpy(verts, sc->verts, sizeof(float) * 3 * sc->n_verts);
  ora_scene tmp = *sc;
  tmp.verts = verts;
  w->verts = verts;
  for (int i = 0; i < sc->n_shapes; ++i) shapeInit(&w->shapes[i], &tmp, i);
  w->child_shape = (int *)malloc(sizeof(int) * (sc->n_children + 1));
  w->child_T = (xf *)malloc(sizeof(xf) * (sc->n_children + 1));
  for (int i = 0; i < sc->n_children; ++i) {
    w->child_shape[i] = sc->child_shape[i];
    w->child_T[i] = xfFrom7(sc->child_T + 7 * i);
  }
  size_t no = sc->n_objects;
  w->obj_off = (int *)malloc(sizeof(int) * (no + 1)); w->obj_cnt = (int *)malloc(sizeof(int) * (no + 1));
  w->obj_compound = (int *)malloc(sizeof(int) * (no + 1)); w->obj_group = (int *)malloc(sizeof(int) * (no + 1));
  for (size_t i = 0; i < no; ++i) {
    w->obj_off[i] = sc->obj_child_off[i]; w->obj_cnt[i] = sc->obj_child_cnt[i];
    w->obj_compound[i] = sc->obj_compound[i]; w->obj_group[i] = sc->obj_group[i];
  }
  w->allowed = (unsigned char *)malloc(no * no + 1);
  memset(w->allowed, 1, no * no + 1);
  w->d = 0.05f;
  w->breaking = (float)(2.001 * (double)w->d);
  return w;
}
void ora_destroy(void *h) {
  world_t *w = (world_t *)h;
  free(w->verts);
  free(w->shapes); free(w->child_shape); free(w->child_T); free(w->obj_off); free(w->obj_cnt);
  free(w->obj_compound); free(w->obj_group); free(w->allowed); free(w);
}
void ora_set_contact_distance(void *h, float d) { /* [bcc:449-460] */
  world_t *w = (world_t *)h;
  w->d = d;
  w->breaking = (float)(2.001 * (double)d);
}
void ora_set_allowed(void *h, const unsigned char *m) {
  world_t *w = (world_t *)h;
  memcpy(w->allowed, m, (size_t)w->n_objects * w->n_objects);
}
long ora_counter(void *h, int which) {
  world_t *w = (world_t *)h;
  switch (which) {
    case 0: return w->pair_evals;
    case 1: return w->aabb_tests;
    case 2: return g_gjk_calls;
    case 3: return g_gjk_iters;
    case 4: return g_epa_calls;
    case 5: return g_epa_iters;
  }
  return -1;
}

/* Engine-defined conservative cull (NOT Bullet's: the DBVT cull is history dependent and, inflated by >= d on both
 * sides, can never remove a pair whose distance is <= d).  Exact local AABB of the shape -> world AABB by
 * |R|*half-extent about the transformed centre -> each side grown by d -> closed-interval overlap.  The HIP
 * engine restates this op for op, so both count the same pair-evals. */
typedef struct { v3 lo, hi; } aabb_t;
static aabb_t shapeAabb(const shape_t *s, const xf *t, float d) {
  v3 he = vadd(vscale(vsub(s->lmax, s->lmin), 0.5f), V(d, d, d));
  v3 lc = vscale(vadd(s->lmax, s->lmin), 0.5f);
  v3 c = xfApply(t, lc);
  v3 a0 = V(fabsf(t->b.r[0].x), fabsf(t->b.r[0].y), fabsf(t->b.r[0].z));
  v3 a1 = V(fabsf(t->b.r[1].x), fabsf(t->b.r[1].y), fabsf(t->b.r[1].z));
  v3 a2 = V(fabsf(t->b.r[2].x), fabsf(t->b.r[2].y), fabsf(t->b.r[2].z));
  v3 e = V(vdot(he, a0), vdot(he, a1), vdot(he, a2));
  aabb_t r = {vsub(c, e), vadd(c, e)};
  return r;
}
static aabb_t aabbUnion(aabb_t a, aabb_t b) {
  aabb_t r = {V(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z)),
              V(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z))};
  return r;
}
static int aabbOverlap(aabb_t a, aabb_t b) {
  return a.lo.x <= b.hi.x && b.lo.x <= a.hi.x && a.lo.y <= b.hi.y && b.lo.y <= a.hi.y && a.lo.z <= b.hi.z && b.lo.z <= a.hi.z;
}

static cvx_t plainCvx(const shape_t *s) {
  cvx_t c;
  memset(&c, 0, sizeof c);
  c.s = s;
  return c;
}

static xf childWorld(const world_t *w, int obj, int child, const xf *objT) {
  /* bare shape: the object's transform; compound: orgTrans*childTrans [cpa:122-125] */
  if (!w->obj_compound[obj]) return *objT;
  return xfMul(objT, &w->child_T[w->obj_off[obj] + child]);
}

/* one narrowphase evaluation with Bullet's A/B roles; returns 1 and fills c when a contact is kept */
static int pairEval(world_t *w, const cvx_t *A, const xf *tA, const cvx_t *B, const xf *tB, ora_contact *c) {
  float md = cvxMargin(A) + cvxMargin(B) + w->breaking; /* [cca:370-373] */
  md *= md;
  gjkout_t o;
  w->pair_evals++;
  if (!A->cast && !B->cast && A->s->kind == ORA_SPHERE && B->s->kind == ORA_SPHERE) {
    /* two plain spheres never reach GJK: the dispatch matrix holds btSphereSphereCollisionAlgorithm for them
     * (btDefaultCollisionConfiguration.cpp:219-222; also for a compound's sphere child, whose algorithm is found by
     * the CHILD's shape type, btCompoundCollisionAlgorithm.cpp:65).  processCollision
     * [btSphereSphereCollisionAlgorithm.cpp:42-89]: no contact at all while the surfaces are apart. */
    v3 diff = vsub(tA->o, tB->o);
    float len = vlen(diff);
    float radius0 = A->s->dims.x * 1.0f, radius1 = B->s->dims.x * 1.0f; /* getRadius(): implicit dims.x * scaling.x */
    if (len > (radius0 + radius1)) return 0;
    o.distance = len - (radius0 + radius1);
    o.normalOnB = V(1, 0, 0);
    if (len > SIMD_EPSILON) o.normalOnB = vdiv(diff, len);
    o.pointOnB = vadd(tB->o, vscale(o.normalOnB, radius1));
    o.valid = 1;
  } else
    gjkPair(A, B, tA, tB, md, &o);
  if (!o.valid) return 0;
  /* btBridgedManifoldResult::addContactPoint [cw:1068-1086]; CollisionCollector [bcc:398-408] */
  if ((double)o.distance > (double)(w->all_pairs ? w->breaking : w->d)) return 0; /* AllVsAll: btManifoldResult::addContactPoint [mr:93-96] */
  v3 pA = vadd(o.pointOnB, vscale(o.normalOnB, o.distance));
  c->dist = o.distance;
  c->n[0] = o.normalOnB.x; c->n[1] = o.normalOnB.y; c->n[2] = o.normalOnB.z;
  c->ptA[0] = pA.x; c->ptA[1] = pA.y; c->ptA[2] = pA.z;
  c->ptB[0] = o.pointOnB.x; c->ptB[1] = o.pointOnB.y; c->ptB[2] = o.pointOnB.z;
  c->ptB1[0] = c->ptB1[1] = c->ptB1[2] = 0;
  c->time = 0;
  c->cctype = ORA_CC_NONE;
  return 1;
}

int ora_links_vs_all(void *h, const double *objT7, const int *q, int nq, int filterMask, ora_contact *out, int cap) {
  world_t *w = (world_t *)h;
  int n = 0;
  xf *T = (xf *)malloc(sizeof(xf) * (w->n_objects + 1));
  for (int i = 0; i < w->n_objects; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    for (int X = 0; X < w->n_objects; ++X) {
      if (X == Q) continue; /* [cw:1132-1133] */
      /* needsCollision [bcc:409-413]; query group is DefaultFilter(1), proxy masks are AllFilter */
      if (!((w->obj_group[X] & (short)filterMask))) continue;
      if (!w->allowed[(size_t)Q * w->n_objects + X]) continue;
      for (int qc = 0; qc < w->obj_cnt[Q]; ++qc) {
        xf tq = childWorld(w, Q, qc, &T[Q]);
        cvx_t cq = plainCvx(&w->shapes[w->child_shape[w->obj_off[Q] + qc]]);
        aabb_t bq = shapeAabb(cq.s, &tq, w->d);
        for (int xc = 0; xc < w->obj_cnt[X]; ++xc) {
          xf tx = childWorld(w, X, xc, &T[X]);
          cvx_t cx = plainCvx(&w->shapes[w->child_shape[w->obj_off[X] + xc]]);
          aabb_t bx = shapeAabb(cx.s, &tx, w->d);
          w->aabb_tests++;
          if (!aabbOverlap(bq, bx)) continue;
          ora_contact c;
          memset(&c, 0, sizeof c);
          int kept;
          /* GJK roles: X compound -> A = X child, B = Q child; else A = Q child, B = X */
          if (w->obj_compound[X]) {
            kept = pairEval(w, &cx, &tx, &cq, &tq, &c);
            c.objA = X; c.childA = xc; c.objB = Q; c.childB = qc;
          } else {
            kept = pairEval(w, &cq, &tq, &cx, &tx, &c);
            c.objA = Q; c.childA = qc; c.objB = X; c.childB = xc;
          }
          if (kept) {
            if (n < cap) out[n] = c;
            ++n;
          }
        }
      }
    }
  }
  free(T);
  return n;
}

/* AllVsAll [bcc:463-498]: performDiscreteCollisionDetection with the near callback of [bcc:415-422], then every point of
 * every persistent manifold, weight 1 / numContacts.  What that amounts to with this configuration:
 *   - every unordered object pair once, (obj0, obj1) = (lower, higher) id = order of addCollisionObject
 *     (btBroadphasePair sorts its proxies by unique id); all proxy masks are AllFilter, so only CanCollide(obj0, obj1)
 *     filters [bcc:418];
 *   - the algorithms, and with them the GJK roles, are those contactTest(obj0) finds against obj1 (a compound's
 *     child is A, the compound of obj1 winning; else obj0 is A);
 *   - every convex pair owns a manifold [cca:300-306] that GJK feeds ONE point (no perturbation iterations in
 *     btDefaultCollisionConfiguration), so numContacts = 1 and weight = 1;
 *   - btManifoldResult::addContactPoint keeps the point up to the manifold's breaking threshold = 2.001 d
 *     (NOT d: there is no CollisionCollector here), stores it in the two OBJECTS' frames and, own manifold,
 *     refreshContactPoints [pm:199-259] then re-derives the world points from those and the distance as
 *     (posA - posB) . normal -- the values AllVsAll reads;
 *   - nothing in the cull loses a pair within 2 d (every box is grown by d, object boxes by another 2.001 d);
 *     a compound child pair whose distance lies in (2 d, 2.001 d] may fail Bullet's child box test [cpa:128-132]; this
 *     restatement keeps such a pair.
 * Writes contacts with A / B in manifold order; returns the count. */
int ora_all_vs_all(void *h, const double *objT7, ora_contact *out, int cap) {
  world_t *w = (world_t *)h;
  int n = 0;
  xf *T = (xf *)malloc(sizeof(xf) * (w->n_objects + 1));
  for (int i = 0; i < w->n_objects; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  w->all_pairs = 1;
  for (int Q = 0; Q < w->n_objects; ++Q) {
    for (int X = Q + 1; X < w->n_objects; ++X) {
      if (!w->allowed[(size_t)Q * w->n_objects + X]) continue;
      for (int qc = 0; qc < w->obj_cnt[Q]; ++qc) {
        xf tq = childWorld(w, Q, qc, &T[Q]);
        cvx_t cq = plainCvx(&w->shapes[w->child_shape[w->obj_off[Q] + qc]]);
        aabb_t bq = shapeAabb(cq.s, &tq, w->breaking);
        for (int xc = 0; xc < w->obj_cnt[X]; ++xc) {
          xf tx = childWorld(w, X, xc, &T[X]);
          cvx_t cx = plainCvx(&w->shapes[w->child_shape[w->obj_off[X] + xc]]);
          aabb_t bx = shapeAabb(cx.s, &tx, w->breaking);
          w->aabb_tests++;
          if (!aabbOverlap(bq, bx)) continue;
          ora_contact c;
          memset(&c, 0, sizeof c);
          int kept;
          const xf *oA, *oB;
          if (w->obj_compound[X]) {
            kept = pairEval(w, &cx, &tx, &cq, &tq, &c);
            c.objA = X; c.childA = xc; c.objB = Q; c.childB = qc;
            oA = &T[X]; oB = &T[Q];
          } else {
            kept = pairEval(w, &cq, &tq, &cx, &tx, &c);
            c.objA = Q; c.childA = qc; c.objB = X; c.childB = xc;
            oA = &T[Q]; oB = &T[X];
          }
          if (!kept) continue;
          /* the manifold walk asks CanCollide again, in the manifold's order of the two objects [bcc:484]; with a
           * one-directional exclusion (ExcludeCollisionPair, bcc:342-346) that is a second condition */
          if (!w->allowed[(size_t)c.objA * w->n_objects + c.objB]) continue;
          /* btManifoldResult::addContactPoint [mr:85-126] -> refreshContactPoints [pm:199-259] */
          v3 nW = V((float)c.n[0], (float)c.n[1], (float)c.n[2]);
          v3 pointA = V((float)c.ptA[0], (float)c.ptA[1], (float)c.ptA[2]);
          v3 pointB = V((float)c.ptB[0], (float)c.ptB[1], (float)c.ptB[2]);
          v3 localA = mulVM(vsub(pointA, oA->o), &oA->b); /* invXform, btTransform.h:218-222 */
          v3 localB = mulVM(vsub(pointB, oB->o), &oB->b);
          v3 posA = xfApply(oA, localA), posB = xfApply(oB, localB);
          float dist = vdot(vsub(posA, posB), nW);
          if (!(dist <= w->breaking)) continue; /* validContactDistance */
          v3 projected = vsub(posA, vscale(nW, dist));
          v3 diff = vsub(posB, projected);
          if (vdot(diff, diff) > w->breaking * w->breaking) continue;
          c.dist = dist;
          c.ptA[0] = posA.x; c.ptA[1] = posA.y; c.ptA[2] = posA.z;
          c.ptB[0] = posB.x; c.ptB[1] = posB.y; c.ptB[2] = posB.z;
          if (n < cap) out[n] = c;
          ++n;
        }
      }
    }
  }
  w->all_pairs = 0;
  free(T);
  return n;
}

/* GetAverageSupport [bcc:98-131] for polyhedral shapes (box has 8 vertices via btBoxShape::getVertex, btBoxShape.h:142-150) */
static void averageSupport(const shape_t *s, v3 nl, float *outsup, v3 *outpt) {
  if (s->kind == ORA_BOX || s->kind == ORA_HULL) {
    v3 sum = V(0, 0, 0);
    float cnt = 0, maxS = -1000;
    const float EPS = 1e-3f;
    int np = s->kind == ORA_BOX ? 8 : s->nv;
    for (int i = 0; i < np; ++i) {
      v3 pt;
      if (s->kind == ORA_BOX) {
        v3 he = V(s->dims.x + s->margin, s->dims.y + s->margin, s->dims.z + s->margin);
        pt = V(he.x * (1 - (i & 1)) - he.x * (i & 1), he.y * (1 - ((i & 2) >> 1)) - he.y * ((i & 2) >> 1),
               he.z * (1 - ((i & 4) >> 2)) - he.z * ((i & 4) >> 2));
      } else
        pt = V(s->v[3 * i], s->v[3 * i + 1], s->v[3 * i + 2]);
      float sup = vdot(pt, nl);
      if (sup > maxS + EPS) { cnt = 1; sum = pt; maxS = sup; }
      else if (sup < maxS - EPS) {}
      else { cnt += 1; sum = vadd(sum, pt); }
    }
    *outsup = maxS;
    *outpt = vdiv(sum, cnt);
  } else {
    /* non-polyhedral: localGetSupportingVertexWithoutMargin (virtual) */
    *outpt = (s->kind == ORA_SPHERE) ? V(0, 0, 0) : cylZSupport(s->dims, nl);
    *outsup = vdot(nl, *outpt);
  }
}

int ora_cast_vs_all(void *h, const double *objT7, const int *q, int nq, const double *T0_7, const double *T1_7,
                    ora_contact *out, int cap) {
  world_t *w = (world_t *)h;
  int n = 0;
  xf *T = (xf *)malloc(sizeof(xf) * (w->n_objects + 1));
  for (int i = 0; i < w->n_objects; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    xf L0 = xfFrom7(T0_7 + 7 * qi), L1 = xfFrom7(T1_7 + 7 * qi);
    for (int qc = 0; qc < w->obj_cnt[Q]; ++qc) {
      /* CheckShapeCast [bcc:974-998] */
      xf tf0 = L0, tf1 = L1;
      if (w->obj_compound[Q]) {
        tf0 = xfMul(&L0, &w->child_T[w->obj_off[Q] + qc]);
        tf1 = xfMul(&L1, &w->child_T[w->obj_off[Q] + qc]);
      }
      cvx_t cq;
      cq.s = &w->shapes[w->child_shape[w->obj_off[Q] + qc]];
      cq.cast = 1;
      cq.t01 = xfInverseTimes(&tf0, &tf1);
      xf tw1 = xfMul(&tf0, &cq.t01); /* m_cow->getWorldTransform() * shape->m_t01 [bcc:904] */
      aabb_t bq = aabbUnion(shapeAabb(cq.s, &tf0, w->d), shapeAabb(cq.s, &tw1, w->d));
      for (int X = 0; X < w->n_objects; ++X) {
        /* the temp object is not in the world, so no self test; mask = KinBodyFilter [bcc:983] */
        if (!(w->obj_group[X] & ORA_GROUP_KINBODY)) continue;
        if (!w->allowed[(size_t)Q * w->n_objects + X]) continue;
        for (int xc = 0; xc < w->obj_cnt[X]; ++xc) {
          xf tx = childWorld(w, X, xc, &T[X]);
          cvx_t cx = plainCvx(&w->shapes[w->child_shape[w->obj_off[X] + xc]]);
          aabb_t bx = shapeAabb(cx.s, &tx, w->d);
          w->aabb_tests++;
          if (!aabbOverlap(bq, bx)) continue;
          if (w->obj_compound[X] && cq.s->kind == ORA_SPHERE && cx.s->kind == ORA_SPHERE) {
            /* The one pair kind whose reference cull is NOT grown by d on either side: the compound's child test
             * [cpa:128-132] takes btSphereShape::getAabb (centre +- radius, untouched by the SHAPE_EXPANSION patch) for
             * the child and CastHullShape::getAabb [bcc:850-856] = the union of the same at both ends of the sweep for
             * the swept sphere.  A swept sphere passing a compound's sphere child at a distance in (0, d] is therefore
             * culled by the reference (every other kind has at least one box grown by d: nothing within d is lost). */
            float rq = cq.s->margin, rx = cx.s->margin;
            v3 lo0 = vsub(tf0.o, V(rq, rq, rq)), hi0 = vadd(tf0.o, V(rq, rq, rq));
            v3 lo1 = vsub(tw1.o, V(rq, rq, rq)), hi1 = vadd(tw1.o, V(rq, rq, rq));
            aabb_t bs = {V(fminf(lo0.x, lo1.x), fminf(lo0.y, lo1.y), fminf(lo0.z, lo1.z)), V(fmaxf(hi0.x, hi1.x), fmaxf(hi0.y, hi1.y), fmaxf(hi0.z, hi1.z))};
            aabb_t bc = {vsub(tx.o, V(rx, rx, rx)), vadd(tx.o, V(rx, rx, rx))};
            if (!aabbOverlap(bs, bc)) {
              w->pair_evals++; /* it survived the engine's own cull: a pair evaluation, which ends here */
              continue;
            }
          }
          ora_contact c;
          memset(&c, 0, sizeof c);
          int kept, castFirst;
          if (w->obj_compound[X]) {
            kept = pairEval(w, &cx, &tx, &cq, &tf0, &c);
            c.objA = X; c.childA = xc; c.objB = Q; c.childB = qc;
            castFirst = 0;
          } else {
            kept = pairEval(w, &cq, &tf0, &cx, &tx, &c);
            c.objA = Q; c.childA = qc; c.objB = X; c.childB = xc;
            castFirst = 1;
          }
          if (!kept) continue;
          /* CastCollisionCollector::addSingleResult [bcc:894-972]; all locals are float */
          v3 nW = V((float)c.n[0], (float)c.n[1], (float)c.n[2]);
          v3 pAw = V((float)c.ptA[0], (float)c.ptA[1], (float)c.ptA[2]);
          v3 pBw = V((float)c.ptB[0], (float)c.ptB[1], (float)c.ptB[2]);
          v3 nFromCast = vscale(nW, -(float)(castFirst ? 1 : -1));
          v3 nl0 = mulVM(nFromCast, &tf0.b);
          v3 nl1 = mulVM(nFromCast, &tw1.b);
          if (castFirst) {
            for (int k = 0; k < 3; ++k) {
              double t = c.ptA[k]; c.ptA[k] = c.ptB[k]; c.ptB[k] = t;
              c.n[k] *= -1;
            }
            int t;
            t = c.objA; c.objA = c.objB; c.objB = t;
            t = c.childA; c.childA = c.childB; c.childB = t;
          }
          v3 pl0, pl1;
          float ls0, ls1;
          averageSupport(cq.s, nl0, &ls0, &pl0);
          v3 pw0 = xfApply(&tf0, pl0);
          averageSupport(cq.s, nl1, &ls1, &pl1);
          v3 pw1 = xfApply(&tw1, pl1);
          float sup0 = vdot(nFromCast, pw0);
          float sup1 = vdot(nFromCast, pw1);
          const float SUPPORT_FUNC_TOLERANCE = .01f;
          if (sup0 - sup1 > SUPPORT_FUNC_TOLERANCE) {
            c.time = 0; c.cctype = ORA_CC_TIME0;
          } else if (sup1 - sup0 > SUPPORT_FUNC_TOLERANCE) {
            c.time = 1; c.cctype = ORA_CC_TIME1;
          } else {
            v3 ptOnCast = castFirst ? pAw : pBw;
            float l0c = vlen(vsub(ptOnCast, pw0)), l1c = vlen(vsub(ptOnCast, pw1));
            c.ptB[0] = pw0.x; c.ptB[1] = pw0.y; c.ptB[2] = pw0.z;
            c.ptB1[0] = pw1.x; c.ptB1[1] = pw1.y; c.ptB1[2] = pw1.z;
            c.cctype = ORA_CC_BETWEEN;
            const float LENGTH_TOLERANCE = .001f;
            if (l0c + l1c < LENGTH_TOLERANCE) c.time = .5f;
            else c.time = l0c / (l0c + l1c);
          }
          if (n < cap) out[n] = c;
          ++n;
        }
      }
    }
  }
  free(T);
  return n;
}

/* ------------------------------------------------------------------------------------------------------ */
/* ContinuousCheckTrajectory [bcc:683-753]: per (link geometry, trajectory segment) the first time of impact  */
/* against the KinBodyFilter objects, by btCollisionWorld::convexSweepTest [cw:994-1056] ->                   */
/* objectQuerySingleInternal [cw:574-622,785-830] -> btContinuousConvexCollision::calcTimeOfImpact            */
/*   [ccc] ext/bullet/src/BulletCollision/NarrowPhaseCollision/btContinuousConvexCollision.cpp:49-241         */
/*   [tu]  ext/bullet/src/LinearMath/btTransformUtil.h:20-140                                                 */
/*   [cs2] ext/bullet/src/BulletCollision/CollisionShapes/btCollisionShape.cpp:32-58                          */
/* The reference's broadphase ray cull [cw:1019-1031] is left out: it is conservative whenever the contact    */
/* distance inflates the tree boxes by >= 1 mm (3.001*d >= 0.001), so it cannot change the result.            */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z, w; } q4;

static float cofac(const m3 *m, int r1, int c1, int r2, int c2) { /* btMatrix3x3.h:722-725 */
  const float *a = &m->r[0].x;
  return a[3 * r1 + c1] * a[3 * r2 + c2] - a[3 * r1 + c2] * a[3 * r2 + c1];
}
static m3 m3Inverse(const m3 *m) { /* btMatrix3x3.h:1019-1029 */
  v3 co = V(cofac(m, 1, 1, 2, 2), cofac(m, 1, 2, 2, 0), cofac(m, 1, 0, 2, 1));
  float det = vdot(m->r[0], co);
  float s = 1.0f / det;
  m3 r;
  r.r[0] = V(co.x * s, cofac(m, 0, 2, 2, 1) * s, cofac(m, 0, 1, 1, 2) * s);
  r.r[1] = V(co.y * s, cofac(m, 0, 0, 2, 2) * s, cofac(m, 0, 2, 1, 0) * s);
  r.r[2] = V(co.z * s, cofac(m, 0, 1, 2, 0) * s, cofac(m, 0, 0, 1, 1) * s);
  return r;
}
static q4 m3GetRotation(const m3 *m) { /* btMatrix3x3.h:448-478, scalar branch */
  const float *e = &m->r[0].x; /* e[3*i+j] */
  float trace = e[0] + e[4] + e[8];
  float temp[4];
  if (trace > 0.0f) {
    float s = sqrtf(trace + 1.0f);
    temp[3] = (s * 0.5f);
    s = 0.5f / s;
    temp[0] = ((e[3 * 2 + 1] - e[3 * 1 + 2]) * s);
    temp[1] = ((e[3 * 0 + 2] - e[3 * 2 + 0]) * s);
    temp[2] = ((e[3 * 1 + 0] - e[3 * 0 + 1]) * s);
  } else {
    int i = e[0] < e[4] ? (e[4] < e[8] ? 2 : 1) : (e[0] < e[8] ? 2 : 0);
    int j = (i + 1) % 3;
    int k = (i + 2) % 3;
    float s = sqrtf(e[3 * i + i] - e[3 * j + j] - e[3 * k + k] + 1.0f);
    temp[i] = s * 0.5f;
    s = 0.5f / s;
    temp[3] = (e[3 * k + j] - e[3 * j + k]) * s;
    temp[j] = (e[3 * j + i] + e[3 * i + j]) * s;
    temp[k] = (e[3 * k + i] + e[3 * i + k]) * s;
  }
  q4 q = {temp[0], temp[1], temp[2], temp[3]};
  return q;
}
static float q4len2(q4 q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; } /* btQuaternion.h:286-317 */
static q4 q4normalize(q4 q) { /* *this /= length() -> *= 1/len, btQuaternion.h:326-380 */
  float s = 1.0f / sqrtf(q4len2(q));
  q4 r = {q.x * s, q.y * s, q.z * s, q.w * s};
  return r;
}
static q4 q4mul(q4 a, q4 b) { /* btQuaternion.h operator*(q1, q2), scalar branch */
  q4 r = {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  return r;
}
static m3 m3FromQuat(q4 q) { /* btMatrix3x3::setRotation, scalar branch [btMatrix3x3.h:204-269] */
  float d = q4len2(q);
  float s = 2.0f / d;
  float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  m3 r;
  r.r[0] = V(1.0f - (yy + zz), xy - wz, xz + wy);
  r.r[1] = V(xy + wz, 1.0f - (xx + zz), yz - wx);
  r.r[2] = V(xz - wy, yz + wx, 1.0f - (xx + yy));
  return r;
}
/* Portable float sin / cos / acos: Taylor series in IEEE double (+, *, /, sqrt only, no contraction), rounded to float.
 * The SAME operation sequence is compiled into the device code (tohc_narrow.h) and the CPU oracle (trajopt_oracle.c), so
 * both produce identical bits; accuracy ~1e-16, i.e. the correctly rounded float in all but ~1e-8 of the cases.
 * sin / cos are only called with |x| <= pi/8 (ANGULAR_MOTION_THRESHOLD, btTransformUtil.h:20,58-61). */
static double ptHorner(const double *c, int n, double x2) {
  double r = c[n - 1];
  for (int k = n - 2; k >= 0; --k) r = c[k] + x2 * r;
  return r;
}
static float ptSin(float xf) {
  const double c[10] = {1.0, -0.16666666666666666, 0.008333333333333333, -0.0001984126984126984, 2.7557319223985893e-06, -2.505210838544172e-08, 1.6059043836821613e-10, -7.647163731819816e-13, 2.8114572543455206e-15, -8.22063524662433e-18};
  const double x = (double)xf;
  return (float)(x * ptHorner(c, 10, x * x));
}
static float ptCos(float xf) {
  const double c[10] = {1.0, -0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05, -2.755731922398589e-07, 2.08767569878681e-09, -1.1470745597729725e-11, 4.779477332387385e-14, -1.5619206968586225e-16};
  const double x = (double)xf;
  return (float)ptHorner(c, 10, x * x);
}
static double ptAsinD(double z) { /* |z| <= ~0.71 */
  const double c[30] = {1.0, 0.16666666666666666, 0.075, 0.044642857142857144, 0.030381944444444444, 0.022372159090909092, 0.017352764423076924, 0.01396484375, 0.011551800896139705, 0.009761609529194078, 0.008390335809616815, 0.0073125258735988454, 0.006447210311889649, 0.005740037670841924, 0.005153309682319905, 0.004660143486915096, 0.004240907093679363, 0.003880964558837669, 0.0035692053938259347, 0.003297059503473485, 0.0030578216492580306, 0.002846178401108942, 0.00265787063820729, 0.0024894486782468836, 0.002338091892111975, 0.0022014739737101384, 0.0020776610325181676, 0.0019650336162772837, 0.0018622264064031275, 0.0017680811205154183};
  return z * ptHorner(c, 30, z * z);
}
static float ptAcos(float xf) { /* xf in [-1, 1] */
  const double PI = 3.141592653589793;
  const double x = (double)xf;
  if (x > 0.5) return (float)(2.0 * ptAsinD(sqrt((1.0 - x) * 0.5)));
  if (x < -0.5) return (float)(PI - 2.0 * ptAsinD(sqrt((1.0 + x) * 0.5)));
  return (float)(PI * 0.5 - ptAsinD(x));
}
/* 0: libm's float routines, what the reference calls (bit-exact against oracle/_ref); 1: the portable routines above,
 * what the device runs (bit-exact against the GPU).  See tohc_narrow.h for why both exist. */
static int g_trig_mode = 0;
void ora_set_trig_mode(int portable) { g_trig_mode = portable; }
static float oSin(float x) { return g_trig_mode ? ptSin(x) : sinf(x); }
static float oCos(float x) { return g_trig_mode ? ptCos(x) : cosf(x); }
static float btAcosf(float x) { /* btScalar.h:391-397 */
  if (x < -1.0f) x = -1.0f;
  if (x > 1.0f) x = 1.0f;
  return g_trig_mode ? ptAcos(x) : acosf(x);
}
/* btTransformUtil::calculateVelocity with timeStep 1 [tu:108-139] */
static void calcVelocity(const xf *t0, const xf *t1, v3 *linVel, v3 *angVel) {
  *linVel = vdiv(vsub(t1->o, t0->o), 1.0f);
  m3 inv = m3Inverse(&t0->b);
  m3 dmat = mulMM(&t1->b, &inv);
  q4 dorn = q4normalize(m3GetRotation(&dmat));
  float angle = 2.0f * btAcosf(dorn.w);
  v3 axis = V(dorn.x, dorn.y, dorn.z);
  float len = vlen2(axis);
  if (len < SIMD_EPSILON * SIMD_EPSILON) axis = V(1, 0, 0);
  else axis = vdiv(axis, sqrtf(len));
  *angVel = vdiv(vscale(axis, angle), 1.0f);
}
/* btTransformUtil::integrateTransform [tu:42-78] */
static xf integrateTransform(const xf *cur, v3 linvel, v3 angvel, float timeStep) {
  const float AMT = 0.5f * (6.283185307179586232f * 0.25f); /* ANGULAR_MOTION_THRESHOLD = 0.5*SIMD_HALF_PI */
  xf out;
  out.o = vadd(cur->o, vscale(linvel, timeStep));
  v3 axis;
  float fAngle = vlen(angvel);
  if (fAngle * timeStep > AMT) fAngle = AMT / timeStep;
  if (fAngle < 0.001f) axis = vscale(angvel, 0.5f * timeStep - (timeStep * timeStep * timeStep) * (0.020833333333f) * fAngle * fAngle);
  else axis = vscale(angvel, oSin(0.5f * fAngle * timeStep) / fAngle);
  q4 dorn = {axis.x, axis.y, axis.z, oCos(fAngle * timeStep * 0.5f)};
  q4 orn0 = m3GetRotation(&cur->b);
  q4 pred = q4normalize(q4mul(dorn, orn0));
  out.b = m3FromQuat(pred);
  return out;
}
/* btCollisionShape::getAngularMotionDisc [cs2:32-58] over getAabb(identity) of each kind: btTransformAabb with the
 * patched SHAPE_EXPANSION [lm btAabbUtil2.h:182-207]; a hull's cached local box still carries the 0.04 construction
 * margin (recalcLocalAabb at addPoint time, btPolyhedralConvexShape.cpp:449-478; setMargin(0) does not recompute it) */
static float angularMotionDisc(const shape_t *s, float d) {
  v3 mn, mx;
  if (s->kind == ORA_SPHERE) { /* btSphereShape::getAabb: centre +- margin, no expansion */
    mn = V(0 - s->margin, 0 - s->margin, 0 - s->margin);
    mx = V(0 + s->margin, 0 + s->margin, 0 + s->margin);
  } else if (s->kind == ORA_HULL) {
    const float CDM = 0.04f;
    v3 lmax = V(s->lmax.x + CDM, s->lmax.y + CDM, s->lmax.z + CDM), lmin = V(s->lmin.x - CDM, s->lmin.y - CDM, s->lmin.z - CDM);
    v3 he = vscale(vsub(lmax, lmin), 0.5f);
    he = vadd(he, vadd(V(s->margin, s->margin, s->margin), V(d, d, d)));
    v3 lc = vscale(vadd(lmax, lmin), 0.5f);
    v3 c = V(1.0f * lc.x + 0.0f * lc.y + 0.0f * lc.z + 0.0f, 0.0f * lc.x + 1.0f * lc.y + 0.0f * lc.z + 0.0f, 0.0f * lc.x + 0.0f * lc.y + 1.0f * lc.z + 0.0f);
    mn = vsub(c, he); mx = vadd(c, he);
  } else { /* box, cylinder: btTransformAabb(halfExtentsWithoutMargin, margin, identity) */
    v3 he = vadd(vadd(s->dims, V(s->margin, s->margin, s->margin)), V(d, d, d));
    mn = vsub(V(0, 0, 0), he); mx = vadd(V(0, 0, 0), he);
  }
  float radius = vlen(vsub(mx, mn)) * 0.5f;
  v3 center = vscale(vadd(mn, mx), 0.5f);
  return radius + vlen(center);
}

/* calcTimeOfImpact [ccc:90-241] with B at rest (fromB == toB); returns 1 on a hit */
static int timeOfImpact(const world_t *w, const shape_t *sA, const xf *fromA, const xf *toA, const shape_t *sB, const xf *tB, float *fraction,
                        v3 *normal, v3 *hitPoint) {
  v3 linVelA, angVelA, linVelB, angVelB;
  calcVelocity(fromA, toA, &linVelA, &angVelA);
  calcVelocity(tB, tB, &linVelB, &angVelB);
  float boundingRadiusA = angularMotionDisc(sA, w->d);
  float boundingRadiusB = angularMotionDisc(sB, w->d);
  float maxAngularProjectedVelocity = vlen(angVelA) * boundingRadiusA + vlen(angVelB) * boundingRadiusB;
  v3 relLinVel = vsub(linVelB, linVelA);
  float relLinVelocLength = vlen(vsub(linVelB, linVelA));
  if ((relLinVelocLength + maxAngularProjectedVelocity) == 0.f) return 0;
  float lambda = 0.f;
  v3 n = V(0, 0, 0), c;
  float lastLambda = lambda;
  int numIter = 0;
  const int maxIter = 64;
  const float radius = 0.001f;
  cvx_t A = plainCvx(sA), B = plainCvx(sB);
  gjkout_t o;
  gjkPair(&A, &B, fromA, tB, BT_LARGE_FLOAT, &o); /* computeClosestPoints [ccc:49-59]: m_maximumDistanceSquared = BT_LARGE_FLOAT */
  if (!o.valid) return 0;
  c = o.pointOnB;
  float dist = o.distance + 0.0f; /* + result.m_allowedPenetration */
  n = o.normalOnB;
  float projectedLinearVelocity = vdot(relLinVel, n);
  if ((projectedLinearVelocity + maxAngularProjectedVelocity) <= SIMD_EPSILON) return 0;
  while (dist > radius) {
    projectedLinearVelocity = vdot(relLinVel, n);
    if ((projectedLinearVelocity + maxAngularProjectedVelocity) <= SIMD_EPSILON) return 0;
    float dLambda = dist / (projectedLinearVelocity + maxAngularProjectedVelocity);
    lambda = lambda + dLambda;
    if (lambda > 1.0f) return 0;
    if (lambda < 0.0f) return 0;
    if (lambda <= lastLambda) return 0;
    lastLambda = lambda;
    xf iA = integrateTransform(fromA, linVelA, angVelA, lambda);
    xf iB = integrateTransform(tB, linVelB, angVelB, lambda);
    gjkPair(&A, &B, &iA, &iB, BT_LARGE_FLOAT, &o);
    if (o.valid) {
      dist = o.distance + 0.0f;
      c = o.pointOnB;
      n = o.normalOnB;
    } else
      return 0;
    numIter++;
    if (numIter > maxIter) return 0;
  }
  *fraction = lambda;
  *normal = n;
  *hitPoint = c;
  return 1;
}

/* one (swept geometry, other geometry) pair of the sweep below, for checking the reference's choice among equal
 * fractions: link Q's geometry qc moves from T0_7 to T1_7 (link poses), object X's geometry xc rests at XT7 */
int ora_sweep_pair(void *h, int Q, int qc, const double *T0_7, const double *T1_7, int X, int xc, const double *XT7, ora_sweep_hit *out) {
  world_t *w = (world_t *)h;
  xf t0 = xfFrom7(T0_7), t1 = xfFrom7(T1_7), tX = xfFrom7(XT7);
  if (w->obj_compound[Q]) {
    t0 = xfMul(&t0, &w->child_T[w->obj_off[Q] + qc]);
    t1 = xfMul(&t1, &w->child_T[w->obj_off[Q] + qc]);
  }
  xf tx = childWorld(w, X, xc, &tX);
  float fr;
  v3 nrm, pt;
  if (!timeOfImpact(w, &w->shapes[w->child_shape[w->obj_off[Q] + qc]], &t0, &t1, &w->shapes[w->child_shape[w->obj_off[X] + xc]], &tx, &fr, &nrm, &pt)) return 0;
  if (!(vlen2(nrm) > 0.0001f)) return 0;
  nrm = vdiv(nrm, vlen(nrm));
  memset(out, 0, sizeof *out);
  out->obj = Q; out->child = qc; out->hitObj = X; out->hitChild = xc; out->fraction = fr;
  out->n[0] = nrm.x; out->n[1] = nrm.y; out->n[2] = nrm.z;
  out->pt[0] = pt.x; out->pt[1] = pt.y; out->pt[2] = pt.z;
  return 1;
}

/* traj7: [nsteps][nq][7] world poses of the query links; the other objects stay at objT7.  One hit at most per
 * (link, geometry, segment): the closest one; ties go to the lowest (object, geometry) -- the reference takes the first
 * in DBVT traversal order, which is history dependent. */
int ora_continuous_check(void *h, const double *objT7, const int *q, int nq, const double *traj7, int nsteps, ora_sweep_hit *out, int cap) {
  world_t *w = (world_t *)h;
  int n = 0;
  xf *T = (xf *)malloc(sizeof(xf) * (w->n_objects + 1));
  for (int i = 0; i < w->n_objects; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    for (int qc = 0; qc < w->obj_cnt[Q]; ++qc) { /* ContinuousCheckShape [bcc:683-705] */
      const shape_t *sq = &w->shapes[w->child_shape[w->obj_off[Q] + qc]];
      for (int seg = 0; seg + 1 < nsteps; ++seg) {
        xf t0 = xfFrom7(traj7 + 7 * ((size_t)seg * nq + qi)), t1 = xfFrom7(traj7 + 7 * ((size_t)(seg + 1) * nq + qi));
        if (w->obj_compound[Q]) { /* rightMultiplyAll [bcc:674-678] */
          t0 = xfMul(&t0, &w->child_T[w->obj_off[Q] + qc]);
          t1 = xfMul(&t1, &w->child_T[w->obj_off[Q] + qc]);
        }
        float closest = 1.0f; /* ConvexResultCallback::m_closestHitFraction */
        ora_sweep_hit hit;
        memset(&hit, 0, sizeof hit);
        int have = 0;
        for (int X = 0; X < w->n_objects; ++X) {
          if (!(w->obj_group[X] & ORA_GROUP_KINBODY)) continue; /* ccc.m_collisionFilterMask = KinBodyFilter [bcc:688] */
          for (int xc = 0; xc < w->obj_cnt[X]; ++xc) {
            if (closest == 0.0f) break; /* btSingleSweepCallback::process [cw:969-971] (per object there; same result) */
            xf tx = childWorld(w, X, xc, &T[X]);
            const shape_t *sx = &w->shapes[w->child_shape[w->obj_off[X] + xc]];
            float fr;
            v3 nrm, pt;
            if (!timeOfImpact(w, sq, &t0, &t1, sx, &tx, &fr, &nrm, &pt)) continue;
            if (!(vlen2(nrm) > 0.0001f)) continue; /* [cw:601] */
            if (!(fr < closest)) continue;           /* [cw:603] */
            nrm = vdiv(nrm, vlen(nrm));              /* castResult.m_normal.normalize() */
            closest = fr;
            have = 1;
            hit.obj = Q; hit.child = qc; hit.seg = seg; hit.hitObj = X; hit.hitChild = xc; hit.fraction = fr;
            hit.n[0] = nrm.x; hit.n[1] = nrm.y; hit.n[2] = nrm.z;
            hit.pt[0] = pt.x; hit.pt[1] = pt.y; hit.pt[2] = pt.z;
          }
        }
        if (have) {
          if (n < cap) out[n] = hit;
          ++n;
        }
      }
    }
  }
  free(T);
  return n;
}

/* ------------------------------------------------------------------------------------------------------ */
/* RayCastCollision [bcc:352-360]: btCollisionWorld::rayTest [cw:916-932] with a ClosestRayResultCallback     */
/* ------------------------------------------------------------------------------------------------------ */
static v3 interp3(v3 a, v3 b, float rt) { /* btVector3::setInterpolate3, scalar branch [btVector3.h:494-497] */
  float s = 1.0f - rt;
  return V(s * a.x + rt * b.x, s * a.y + rt * b.y, s * a.z + rt * b.z);
}
/* btSubsimplexConvexCast::calcTimeOfImpact [btSubSimplexConvexCast.cpp:38-156] as rayTestSingleInternal calls it
 * [cw:279-301]: convexA is btSphereShape(0) with margin 0 -- its support vertex is (0,0,0) in every direction --,
 * fromA / toA are identity rotations at the ray's ends [cw:856-859], B rests (fromB == toB == its world transform) and
 * answers through the VIRTUAL localGetSupportingVertex (margin 0 for box / hull / cylinder, the radius for a sphere). */
static int rayConvex(const shape_t *sB, const xf *tB, v3 from, v3 to, float *fraction, v3 *normal) {
  vss_t sx;
  vssReset(&sx);
  v3 linVelA = vsub(to, from), linVelB = vsub(tB->o, tB->o);
  float lambda = 0.0f;
  v3 oA = from;
  xf iB = *tB; /* interpolatedTransB: only its origin is ever rewritten */
  v3 r = vsub(linVelA, linVelB);
  const v3 zero = V(0, 0, 0);
  v3 supA = vadd(zero, oA);
  v3 supB = xfApply(tB, supPlainVirtual(sB, mulVM(r, &tB->b)));
  v3 v = vsub(supA, supB);
  int maxIter = 32; /* MAX_ITERATIONS, single precision */
  v3 n = V(0, 0, 0);
  float dist2 = vlen2(v);
  const float epsilon = 0.0001f;
  while ((dist2 > epsilon) && maxIter--) {
    supA = vadd(zero, oA);
    supB = xfApply(&iB, supPlainVirtual(sB, mulVM(v, &iB.b)));
    v3 w = vsub(supA, supB);
    float VdotW = vdot(v, w);
    if (lambda > 1.0f) return 0;
    if (VdotW > 0.0f) {
      float VdotR = vdot(v, r);
      if (VdotR >= -(SIMD_EPSILON * SIMD_EPSILON)) return 0;
      lambda = lambda - VdotW / VdotR;
      oA = interp3(from, to, lambda);
      iB.o = interp3(tB->o, tB->o, lambda); /* s*o + rt*o: not always o again in fp32 */
      n = v;
    }
    if (!vssInSimplex(&sx, w)) vssAdd(&sx, w, supA, supB);
    int ok = vssUpdate(&sx); /* closest(v): v = m_cachedV whatever the outcome [vss.cpp:263-268] */
    v = sx.cV;
    dist2 = ok ? vlen2(v) : 0.0f;
  }
  *fraction = lambda;
  *normal = (vlen2(n) >= (SIMD_EPSILON * SIMD_EPSILON)) ? vdiv(n, vlen(n)) : V(0, 0, 0);
  if (vdot(*normal, r) >= -0.0f) return 0; /* -result.m_allowedPenetration */
  return 1;
}

/* closest[i] = ClosestRayResultCallback::m_closestHitFraction after rayTest(from_i, to_i): 1 when nothing was hit
 * (RayCastCollision returns hasHit(), i.e. closest < 1).  Every object takes part: the callback's group / mask are
 * DefaultFilter / AllFilter and the proxies' masks are AllFilter [cw:RayResultCallback::needsCollision]; the allowed-
 * pair matrix is not consulted.  No cull: Bullet's DBVT ray cull works on object boxes grown by 3.001 d plus history
 * dependent fat margins.  It matters only for the rare hit the cast reports at a point away from the shape (it keeps
 * simplex vertices computed at earlier lambdas): measured against the reference's rayTest on 2 x 800 000 rays, many of
 * them grazing, 0 rays differ at d = 0.05 and 18 at d = 0, where the reference's tight boxes cull such a hit. */
int ora_ray_cast(void *h, const double *objT7, const double *from3, const double *to3, int n_rays, float *closest) {
  world_t *w = (world_t *)h;
  int hits = 0;
  xf *T = (xf *)malloc(sizeof(xf) * (w->n_objects + 1));
  for (int i = 0; i < w->n_objects; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int i = 0; i < n_rays; ++i) {
    v3 from = V((float)from3[3 * i], (float)from3[3 * i + 1], (float)from3[3 * i + 2]); /* toBt [bcc:66-68] */
    v3 to = V((float)to3[3 * i], (float)to3[3 * i + 1], (float)to3[3 * i + 2]);
    float cl = 1.0f;
    for (int X = 0; X < w->n_objects && cl != 0.0f; ++X) /* btSingleRayCallback::process stops at 0 [cw:881-882] */
      for (int xc = 0; xc < w->obj_cnt[X]; ++xc) {
        xf tx = childWorld(w, X, xc, &T[X]);
        float fr;
        v3 nrm;
        if (!rayConvex(&w->shapes[w->child_shape[w->obj_off[X] + xc]], &tx, from, to, &fr, &nrm)) continue;
        if (!(vlen2(nrm) > 0.0001f)) continue; /* [cw:304] */
        if (fr < cl) cl = fr;                  /* [cw:306], ClosestRayResultCallback::addSingleResult */
      }
    closest[i] = cl;
    hits += cl < 1.0f;
  }
  free(T);
  return hits;
}

/* ------------------------------------------------------------------------------------------------------ */
/* FK, point Jacobian and contact -> affine expression (fp64)                                                */
/*   [cs:12-23,83-133] delegate to OpenRAVE, which is absent: conventions are documented in oracle_api.h.   */
/*   [ct:37-80] CollisionsToDistanceExpressions.                                                             */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { double w, x, y, z; } quat;
typedef struct { double x, y, z; } d3;
typedef struct { quat q; d3 t; } dxf;

static d3 D(double x, double y, double z) { d3 r = {x, y, z}; return r; }
static d3 dadd(d3 a, d3 b) { return D(a.x + b.x, a.y + b.y, a.z + b.z); }
static d3 dsub(d3 a, d3 b) { return D(a.x - b.x, a.y - b.y, a.z - b.z); }
static d3 dscale(d3 a, double s) { return D(a.x * s, a.y * s, a.z * s); }
static double ddot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static d3 dcross(d3 a, d3 b) { return D(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static quat qmul(quat a, quat b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
static quat qnormalize(quat q) {
  double s = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  quat r = {q.w * s, q.x * s, q.y * s, q.z * s};
  return r;
}
static d3 qrot(quat q, d3 v) {
  d3 u = D(q.x, q.y, q.z);
  d3 t = dscale(dcross(u, v), 2.0);
  return dadd(dadd(v, dscale(t, q.w)), dcross(u, t));
}
static dxf dxfMul(dxf a, dxf b) {
  dxf r;
  r.q = qnormalize(qmul(a.q, b.q));
  r.t = dadd(qrot(a.q, b.t), a.t);
  return r;
}
static dxf dxfFrom7(const double *t) {
  dxf r = {{t[0], t[1], t[2], t[3]}, {t[4], t[5], t[6]}};
  return r;
}
static void dxfTo7(dxf a, double *t) {
  t[0] = a.q.w; t[1] = a.q.x; t[2] = a.q.y; t[3] = a.q.z; t[4] = a.t.x; t[5] = a.t.y; t[6] = a.t.z;
}

int ora_robot_dof(const ora_robot *rb) { /* RobotAndDOF::GetDOF [cs:80-82] */
  return rb->n_joint_inds + ((rb->affine & 1) != 0) + ((rb->affine & 2) != 0) + ((rb->affine & 4) != 0) + ((rb->affine & 8) != 0);
}

/* RobotAndDOF::SetDOFValues [cs:12-23] followed by link->GetTransform() for every link */
void ora_fk(const ora_robot *rb, const double *dofs, double *linkT7) {
  double *q = (double *)malloc(sizeof(double) * (rb->n_dof + 1));
  for (int i = 0; i < rb->n_dof; ++i) q[i] = rb->q_robot[i];
  for (int i = 0; i < rb->n_joint_inds; ++i) q[rb->joint_inds[i]] = dofs[i];
  dxf base = dxfFrom7(rb->base_T);
  if (rb->affine) {
    const double *a = dofs + rb->n_joint_inds;
    base.q.w = 1; base.q.x = base.q.y = base.q.z = 0;
    base.t = D(0, 0, 0);
    if (rb->affine & ORA_DOF_X) base.t.x = *a++;
    if (rb->affine & ORA_DOF_Y) base.t.y = *a++;
    if (rb->affine & ORA_DOF_Z) base.t.z = *a++;
    if (rb->affine & ORA_DOF_ROTAXIS) {
      double h = 0.5 * (*a++), s = sin(h);
      base.q.w = cos(h);
      base.q.x = rb->rot_axis[0] * s; base.q.y = rb->rot_axis[1] * s; base.q.z = rb->rot_axis[2] * s;
    }
  }
  dxf *T = (dxf *)malloc(sizeof(dxf) * (rb->n_links + 1));
  for (int i = 0; i < rb->n_links; ++i) {
    dxf parent = rb->parent[i] < 0 ? base : T[rb->parent[i]];
    dxf t = dxfMul(parent, dxfFrom7(rb->T_origin + 7 * i));
    if (rb->jtype[i] != ORA_J_FIXED) {
      double v = rb->offs[i];
      if (rb->dof[i] >= 0) v = rb->mult[i] * q[rb->dof[i]] + rb->offs[i];
      d3 ax = D(rb->axis[3 * i], rb->axis[3 * i + 1], rb->axis[3 * i + 2]);
      dxf m = {{1, 0, 0, 0}, {0, 0, 0}};
      if (rb->jtype[i] == ORA_J_REVOLUTE) {
        double h = 0.5 * v, s = sin(h);
        m.q.w = cos(h); m.q.x = ax.x * s; m.q.y = ax.y * s; m.q.z = ax.z * s;
      } else {
        m.t = dscale(ax, v);
      }
      t = dxfMul(t, m);
    }
    T[i] = t;
    dxfTo7(t, linkT7 + 7 * i);
  }
  free(T);
  free(q);
}

/* RobotAndDOF::PositionJacobian [cs:83-89] -> CalculateActiveJacobian: 3 x K, row major.
 * linkT7 are the link transforms at the configuration the Jacobian is taken at. */
void ora_position_jacobian(const ora_robot *rb, const double *linkT7, const double *dofs, int link, const double *pt, double *J) {
  int K = ora_robot_dof(rb);
  for (int i = 0; i < 3 * K; ++i) J[i] = 0;
  d3 p = D(pt[0], pt[1], pt[2]);
  for (int i = link; i >= 0; i = rb->parent[i]) {
    if (rb->jtype[i] == ORA_J_FIXED || rb->dof[i] < 0) continue;
    int k = -1;
    for (int a = 0; a < rb->n_joint_inds; ++a)
      if (rb->joint_inds[a] == rb->dof[i]) { k = a; break; }
    if (k < 0) continue;
    dxf t = dxfFrom7(linkT7 + 7 * i);
    d3 z = qrot(t.q, D(rb->axis[3 * i], rb->axis[3 * i + 1], rb->axis[3 * i + 2]));
    d3 col = rb->jtype[i] == ORA_J_REVOLUTE ? dcross(z, dsub(p, t.t)) : z;
    J[0 * K + k] += rb->mult[i] * col.x;
    J[1 * K + k] += rb->mult[i] * col.y;
    J[2 * K + k] += rb->mult[i] * col.z;
  }
  int k = rb->n_joint_inds;
  if (rb->affine & ORA_DOF_X) { J[0 * K + k] = 1; ++k; }
  if (rb->affine & ORA_DOF_Y) { J[1 * K + k] = 1; ++k; }
  if (rb->affine & ORA_DOF_Z) { J[2 * K + k] = 1; ++k; }
  if (rb->affine & ORA_DOF_ROTAXIS) {
    /* base translation = the affine x,y,z values (unset components are 0, cs:14-16) */
    const double *a = dofs + rb->n_joint_inds;
    d3 bt = D(0, 0, 0);
    if (rb->affine & ORA_DOF_X) bt.x = *a++;
    if (rb->affine & ORA_DOF_Y) bt.y = *a++;
    if (rb->affine & ORA_DOF_Z) bt.z = *a++;
    d3 col = dcross(D(rb->rot_axis[0], rb->rot_axis[1], rb->rot_axis[2]), dsub(p, bt));
    J[0 * K + k] = col.x; J[1 * K + k] = col.y; J[2 * K + k] = col.z;
    ++k;
  }
}

/* RobotAndDOF::DoesAffect [cs:97-101] + trajopt::DoesAffect (rave_utils.cpp:69-74) */
int ora_link_affected(const ora_robot *rb, int link) {
  if (rb->affine > 0) return 1;
  if (rb->dof[link] == ORA_DOF_GRABBED) return 1; /* a grabbed body's link: listed whatever moves it [cs:120-131] */
  for (int i = link; i >= 0; i = rb->parent[i]) {
    if (rb->jtype[i] == ORA_J_FIXED || rb->dof[i] < 0) continue;
    for (int a = 0; a < rb->n_joint_inds; ++a)
      if (rb->joint_inds[a] == rb->dof[i]) return 1;
  }
  return 0;
}

/* GetAffectedLinks(links, only_with_geom=true, inds) [cs:110-118]: query objects and their Jacobian link */
static int affectedObjs(const ora_robot *rb, int *objs, int *links) {
  int n = 0;
  for (int i = 0; i < rb->n_links; ++i)
    if (rb->link_obj[i] >= 0 && ora_link_affected(rb, i)) { objs[n] = rb->link_obj[i]; links[n] = i; ++n; }
  return n;
}

/* one side of CollisionsToDistanceExpressions [ct:44-56]: grad += sign * n^T J(link, pt); returns grad.q */
static void accumSide(const ora_robot *rb, const double *linkT7, const double *dofs, int link, const double *pt,
                      const double *n, double sign, double *grad, double *J) {
  int K = ora_robot_dof(rb);
  ora_position_jacobian(rb, linkT7, dofs, link, pt, J);
  for (int k = 0; k < K; ++k) grad[k] += sign * (n[0] * J[k] + n[1] * J[K + k] + n[2] * J[2 * K + k]);
}

static int objLink(const int *objs, const int *links, int n, int obj) {
  for (int i = 0; i < n; ++i)
    if (objs[i] == obj) return links[i];
  return -1;
}

static void linearizeAt(const ora_robot *rb, const double *linkT7, const double *dofs, const int *qobjs, const int *qlinks,
                        int nq, const ora_contact *c, int timestep1, double *konst, double *grad, double *J) {
  int K = ora_robot_dof(rb);
  for (int k = 0; k < K; ++k) grad[k] = 0;
  int la = objLink(qobjs, qlinks, nq, c->objA), lb = objLink(qobjs, qlinks, nq, c->objB);
  if (la >= 0) accumSide(rb, linkT7, dofs, la, c->ptA, c->n, 1.0, grad, J);
  if (lb >= 0) accumSide(rb, linkT7, dofs, lb, (timestep1 && c->cctype == ORA_CC_BETWEEN) ? c->ptB1 : c->ptB, c->n, -1.0, grad, J);
  double gq = 0;
  for (int k = 0; k < K; ++k) gq += grad[k] * dofs[k];
  *konst = c->dist - gq;
}

/* SingleTimestepCollisionEvaluator::CalcDistExpressions for every timestep [ct:118-136].
 * static_T7: world transforms of all objects (robot-link objects are overwritten by FK).  grads: [cap][2][K]. */
int ora_linearize_discrete(void *h, const ora_robot *rb, const double *static_T7, const double *dofs, int T, int filterMask,
                           ora_lin *out, double *grads, int cap) {
  world_t *w = (world_t *)h;
  int K = ora_robot_dof(rb), n = 0;
  int *qobjs = (int *)malloc(sizeof(int) * (rb->n_links + 1)), *qlinks = (int *)malloc(sizeof(int) * (rb->n_links + 1));
  int nq = affectedObjs(rb, qobjs, qlinks);
  double *objT = (double *)malloc(sizeof(double) * 7 * (w->n_objects + 1));
  double *linkT = (double *)malloc(sizeof(double) * 7 * (rb->n_links + 1));
  double *J = (double *)malloc(sizeof(double) * 3 * (K + 1));
  ora_contact *cbuf = (ora_contact *)malloc(sizeof(ora_contact) * (cap + 1));
  for (int t = 0; t < T; ++t) {
    const double *q = dofs + (size_t)t * K;
    memcpy(objT, static_T7, sizeof(double) * 7 * w->n_objects);
    ora_fk(rb, q, linkT);
    for (int i = 0; i < rb->n_links; ++i)
      if (rb->link_obj[i] >= 0) memcpy(objT + 7 * rb->link_obj[i], linkT + 7 * i, sizeof(double) * 7);
    int nc = ora_links_vs_all(h, objT, qobjs, nq, filterMask, cbuf, cap);
    for (int i = 0; i < nc && i < cap; ++i) {
      if (n < cap) {
        out[n].t = t;
        out[n].c = cbuf[i];
        out[n].const1 = 0;
        linearizeAt(rb, linkT, q, qobjs, qlinks, nq, &cbuf[i], 0, &out[n].const0, grads + (size_t)n * 2 * K, J);
        for (int k = 0; k < K; ++k) grads[(size_t)n * 2 * K + K + k] = 0;
      }
      ++n;
    }
  }
  free(qobjs); free(qlinks); free(objT); free(linkT); free(J); free(cbuf);
  return n;
}

/* CastCollisionEvaluator::CalcDistExpressions for every pair (t, t+gap) [ct:156-168, 64-80] */
int ora_linearize_cast(void *h, const ora_robot *rb, const double *static_T7, const double *dofs, int T, int gap, ora_lin *out,
                       double *grads, int cap) {
  world_t *w = (world_t *)h;
  int K = ora_robot_dof(rb), n = 0;
  int *qobjs = (int *)malloc(sizeof(int) * (rb->n_links + 1)), *qlinks = (int *)malloc(sizeof(int) * (rb->n_links + 1));
  int nq = affectedObjs(rb, qobjs, qlinks);
  double *objT = (double *)malloc(sizeof(double) * 7 * (w->n_objects + 1));
  double *linkT0 = (double *)malloc(sizeof(double) * 7 * (rb->n_links + 1));
  double *linkT1 = (double *)malloc(sizeof(double) * 7 * (rb->n_links + 1));
  double *T0 = (double *)malloc(sizeof(double) * 7 * (nq + 1)), *T1 = (double *)malloc(sizeof(double) * 7 * (nq + 1));
  double *J = (double *)malloc(sizeof(double) * 3 * (K + 1));
  ora_contact *cbuf = (ora_contact *)malloc(sizeof(ora_contact) * (cap + 1));
  for (int t = 0; t + gap < T; ++t) {
    const double *q0 = dofs + (size_t)t * K, *q1 = dofs + (size_t)(t + gap) * K;
    ora_fk(rb, q0, linkT0);
    ora_fk(rb, q1, linkT1);
    memcpy(objT, static_T7, sizeof(double) * 7 * w->n_objects);
    for (int i = 0; i < rb->n_links; ++i)
      if (rb->link_obj[i] >= 0) memcpy(objT + 7 * rb->link_obj[i], linkT0 + 7 * i, sizeof(double) * 7);
    for (int i = 0; i < nq; ++i) {
      memcpy(T0 + 7 * i, linkT0 + 7 * qlinks[i], sizeof(double) * 7);
      memcpy(T1 + 7 * i, linkT1 + 7 * qlinks[i], sizeof(double) * 7);
    }
    int nc = ora_cast_vs_all(h, objT, qobjs, nq, T0, T1, cbuf, cap);
    for (int i = 0; i < nc && i < cap; ++i) {
      if (n < cap) {
        out[n].t = t;
        out[n].c = cbuf[i];
        linearizeAt(rb, linkT0, q0, qobjs, qlinks, nq, &cbuf[i], 0, &out[n].const0, grads + (size_t)n * 2 * K, J);
        linearizeAt(rb, linkT1, q1, qobjs, qlinks, nq, &cbuf[i], 1, &out[n].const1, grads + (size_t)n * 2 * K + K, J);
      }
      ++n;
    }
  }
  free(qobjs); free(qlinks); free(objT); free(linkT0); free(linkT1); free(T0); free(T1); free(J); free(cbuf);
  return n;
}
