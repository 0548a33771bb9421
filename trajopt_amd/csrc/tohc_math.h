// tohc_math.h -- fp32 vector algebra in btVector3 / btMatrix3x3 operation order, and the fp64 quaternion algebra of
// the FK model.  Shared by every kernel.  Compiled with -ffp-contract=off: each multiply and add rounds on its own,
// as in the reference build, otherwise borderline contacts flip.
//
// Reference: ext/bullet/src/LinearMath/{btVector3.h:229-247,380-384,838, btMatrix3x3.h:204-269,611-622,1074-1084,1187,1270,
// btTransform.h:94-97,228-242}.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TOHC_HD __host__ __device__ __forceinline__
#define TOHC_DEV __device__ __forceinline__
#else
#include <cmath>
#define TOHC_HD inline
#define TOHC_DEV inline
#endif

namespace tohc {

struct V3 { float x, y, z; };
struct alignas(16) F4 { float x, y, z, w; };  // one hull vertex as a 16-byte record (lane-per-pair kernels read vertices from memory)
struct M3 { V3 r[3]; };
struct Xf { M3 b; V3 o; };

TOHC_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
TOHC_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
TOHC_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
TOHC_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
TOHC_HD V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
TOHC_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TOHC_HD float len2(V3 a) { return dot(a, a); }
TOHC_HD float len(V3 a) { return sqrtf(len2(a)); }
TOHC_HD V3 vdiv(V3 a, float s) { return a * (1.0f / s); }
TOHC_HD V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TOHC_HD bool veq(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
TOHC_HD V3 mulMV(const M3 &m, V3 v) { return mk(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)); }
TOHC_HD V3 mulVM(V3 v, const M3 &m) {
  return mk(m.r[0].x * v.x + m.r[1].x * v.y + m.r[2].x * v.z, m.r[0].y * v.x + m.r[1].y * v.y + m.r[2].y * v.z,
            m.r[0].z * v.x + m.r[1].z * v.y + m.r[2].z * v.z);
}
TOHC_HD M3 mulMM(const M3 &a, const M3 &b) {
  M3 r;
  for (int i = 0; i < 3; ++i) r.r[i] = mulVM(a.r[i], b);
  return r;
}
TOHC_HD M3 transposeTimes(const M3 &a, const M3 &m) {
  M3 r;
  const V3 *e = a.r, *n = m.r;
  r.r[0] = mk(e[0].x * n[0].x + e[1].x * n[1].x + e[2].x * n[2].x, e[0].x * n[0].y + e[1].x * n[1].y + e[2].x * n[2].y,
              e[0].x * n[0].z + e[1].x * n[1].z + e[2].x * n[2].z);
  r.r[1] = mk(e[0].y * n[0].x + e[1].y * n[1].x + e[2].y * n[2].x, e[0].y * n[0].y + e[1].y * n[1].y + e[2].y * n[2].y,
              e[0].y * n[0].z + e[1].y * n[1].z + e[2].y * n[2].z);
  r.r[2] = mk(e[0].z * n[0].x + e[1].z * n[1].x + e[2].z * n[2].x, e[0].z * n[0].y + e[1].z * n[1].y + e[2].z * n[2].y,
              e[0].z * n[0].z + e[1].z * n[1].z + e[2].z * n[2].z);
  return r;
}
TOHC_HD V3 xfApply(const Xf &t, V3 x) { return mulMV(t.b, x) + t.o; }
TOHC_HD Xf xfMul(const Xf &a, const Xf &b) {
  Xf r;
  r.b = mulMM(a.b, b.b);
  r.o = xfApply(a, b.o);
  return r;
}
TOHC_HD Xf xfInverseTimes(const Xf &a, const Xf &t) {
  Xf r;
  V3 v = t.o - a.o;
  r.b = transposeTimes(a.b, t.b);
  r.o = mulVM(v, a.b);
  return r;
}
// toBt(OR::Transform) (bullet_collision_checker.cpp:72-83) + btMatrix3x3::setRotation scalar branch
TOHC_HD Xf xfFrom7(const double *t) {
  float qx = (float)t[1], qy = (float)t[2], qz = (float)t[3], qw = (float)t[0];
  float d = qx * qx + qy * qy + qz * qz + qw * qw;
  float s = 2.0f / d;
  float xs = qx * s, ys = qy * s, zs = qz * s;
  float wx = qw * xs, wy = qw * ys, wz = qw * zs;
  float xx = qx * xs, xy = qx * ys, xz = qx * zs;
  float yy = qy * ys, yz = qy * zs, zz = qz * zs;
  Xf r;
  r.b.r[0] = mk(1.0f - (yy + zz), xy - wz, xz + wy);
  r.b.r[1] = mk(xy + wz, 1.0f - (xx + zz), yz - wx);
  r.b.r[2] = mk(xz - wy, yz + wx, 1.0f - (xx + yy));
  r.o = mk((float)t[4], (float)t[5], (float)t[6]);
  return r;
}

// conservative cull box: exact local AABB -> world AABB by |R|*half-extent about the transformed centre, grown by d
struct Aabb { V3 lo, hi; };
TOHC_HD Aabb shapeAabb(V3 lmin, V3 lmax, const Xf &t, float d) {
  V3 he = (lmax - lmin) * 0.5f + mk(d, d, d);
  V3 lc = (lmax + lmin) * 0.5f;
  V3 c = xfApply(t, lc);
  V3 a0 = mk(fabsf(t.b.r[0].x), fabsf(t.b.r[0].y), fabsf(t.b.r[0].z));
  V3 a1 = mk(fabsf(t.b.r[1].x), fabsf(t.b.r[1].y), fabsf(t.b.r[1].z));
  V3 a2 = mk(fabsf(t.b.r[2].x), fabsf(t.b.r[2].y), fabsf(t.b.r[2].z));
  V3 e = mk(dot(he, a0), dot(he, a1), dot(he, a2));
  Aabb r;
  r.lo = c - e;
  r.hi = c + e;
  return r;
}
TOHC_HD Aabb aabbUnion(Aabb a, Aabb b) {
  Aabb r;
  r.lo = mk(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z));
  r.hi = mk(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z));
  return r;
}
TOHC_HD bool aabbOverlap(const Aabb &a, const Aabb &b) {
  return a.lo.x <= b.hi.x && b.lo.x <= a.hi.x && a.lo.y <= b.hi.y && b.lo.y <= a.hi.y && a.lo.z <= b.hi.z && b.lo.z <= a.hi.z;
}

// ---- fp64 FK algebra (conventions: DESIGN.md "FK model"; OpenRAVE itself is absent) ------------------------------
struct Qd { double w, x, y, z; };
struct D3 { double x, y, z; };
struct Dxf { Qd q; D3 t; };
TOHC_HD D3 mkd(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
TOHC_HD D3 operator+(D3 a, D3 b) { return mkd(a.x + b.x, a.y + b.y, a.z + b.z); }
TOHC_HD D3 operator-(D3 a, D3 b) { return mkd(a.x - b.x, a.y - b.y, a.z - b.z); }
TOHC_HD D3 operator*(D3 a, double s) { return mkd(a.x * s, a.y * s, a.z * s); }
TOHC_HD D3 cross(D3 a, D3 b) { return mkd(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TOHC_HD Qd qmul(Qd a, Qd b) {
  Qd r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
TOHC_HD Qd qnormalize(Qd q) {
  double s = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  Qd r;
  r.w = q.w * s; r.x = q.x * s; r.y = q.y * s; r.z = q.z * s;
  return r;
}
TOHC_HD D3 qrot(Qd q, D3 v) {
  D3 u = mkd(q.x, q.y, q.z);
  D3 t = cross(u, v) * 2.0;
  return (v + t * q.w) + cross(u, t);
}
TOHC_HD Dxf dxfMul(const Dxf &a, const Dxf &b) {
  Dxf r;
  r.q = qnormalize(qmul(a.q, b.q));
  r.t = qrot(a.q, b.t) + a.t;
  return r;
}
TOHC_HD Dxf dxfFrom7(const double *t) {
  Dxf r;
  r.q.w = t[0]; r.q.x = t[1]; r.q.y = t[2]; r.q.z = t[3];
  r.t = mkd(t[4], t[5], t[6]);
  return r;
}
TOHC_HD void dxfTo7(const Dxf &a, double *t) {
  t[0] = a.q.w; t[1] = a.q.x; t[2] = a.q.y; t[3] = a.q.z; t[4] = a.t.x; t[5] = a.t.y; t[6] = a.t.z;
}

}  // namespace tohc
