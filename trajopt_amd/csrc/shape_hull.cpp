// shape_hull.cpp -- tohc_shape_hull: the reduction the reference applies to a triangle mesh of >= 50 vertices before it
// becomes a convex hull shape (createShapePrimitive, src/trajopt/bullet_collision_checker.cpp:174-203):
//
//   btShapeHull(&btConvexTriangleMeshShape).buildHull()          ext/bullet/src/BulletCollision/CollisionShapes/btShapeHull.cpp:38-100
//     42 fixed directions -> the mesh's support vertex in each    btShapeHull.cpp:121-169, btConvexTriangleMeshShape.cpp:32-91
//     HullLibrary::CreateConvexHull(QF_TRIANGLES) of those points ext/bullet/src/LinearMath/btConvexHull.cpp:691-812
//       CleanupVertices (normalise to the unit box, merge points closer than 0.001, scale back)   :855-1112
//       calchull: Melax's incremental hull -- FindSimplex, extrude the triangle with the highest rise    :482-644
//       BringOutYourDead: vertices re-numbered in the order the triangle list first names them    :1114-1167
//
// Host set-up code, run once per mesh; fp32 throughout in the reference's operation order (tohc_math.h, compiled with
// -ffp-contract=off), so that the points -- values AND order, which decides ties between support vertices later -- are
// the ones the reference's btConvexHullShape holds.  Checked bit for bit against the reference's own Bullet
// (oracle/_ref, ref_shape_hull) in tests/test_oracle_cpu.py and against tests/golden/shape_hull.npz.
#include <float.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "../../include/trajopt_hipcc.h"
#include "tohc_math.h"

namespace {
using namespace tohc;

// btShapeHull::getUnitSpherePoints(): an icosahedron subdivided once, as printed there (six decimals, not unit length)
const float kDirections[42][3] = {
    {0.000000f, -0.000000f, -1.000000f}, {0.723608f, -0.525725f, -0.447219f}, {-0.276388f, -0.850649f, -0.447219f},
    {-0.894426f, -0.000000f, -0.447216f}, {-0.276388f, 0.850649f, -0.447220f}, {0.723608f, 0.525725f, -0.447219f},
    {0.276388f, -0.850649f, 0.447220f}, {-0.723608f, -0.525725f, 0.447219f}, {-0.723608f, 0.525725f, 0.447219f},
    {0.276388f, 0.850649f, 0.447219f}, {0.894426f, 0.000000f, 0.447216f}, {-0.000000f, 0.000000f, 1.000000f},
    {0.425323f, -0.309011f, -0.850654f}, {-0.162456f, -0.499995f, -0.850654f}, {0.262869f, -0.809012f, -0.525738f},
    {0.425323f, 0.309011f, -0.850654f}, {0.850648f, -0.000000f, -0.525736f}, {-0.525730f, -0.000000f, -0.850652f},
    {-0.688190f, -0.499997f, -0.525736f}, {-0.162456f, 0.499995f, -0.850654f}, {-0.688190f, 0.499997f, -0.525736f},
    {0.262869f, 0.809012f, -0.525738f}, {0.951058f, 0.309013f, 0.000000f}, {0.951058f, -0.309013f, 0.000000f},
    {0.587786f, -0.809017f, 0.000000f}, {0.000000f, -1.000000f, 0.000000f}, {-0.587786f, -0.809017f, 0.000000f},
    {-0.951058f, -0.309013f, -0.000000f}, {-0.951058f, 0.309013f, -0.000000f}, {-0.587786f, 0.809017f, -0.000000f},
    {-0.000000f, 1.000000f, -0.000000f}, {0.587786f, 0.809017f, -0.000000f}, {0.688190f, -0.499997f, 0.525736f},
    {-0.262869f, -0.809012f, 0.525738f}, {-0.850648f, 0.000000f, 0.525736f}, {-0.262869f, 0.809012f, 0.525738f},
    {0.688190f, 0.499997f, 0.525736f}, {0.525730f, 0.000000f, 0.850652f}, {0.162456f, -0.499995f, 0.850654f},
    {-0.425323f, -0.309011f, 0.850654f}, {-0.425323f, 0.309011f, 0.850654f}, {0.162456f, 0.499995f, 0.850654f}};

// btConvexTriangleMeshShape::localGetSupportingVertex with margin 0: the direction is normalised, every corner of every
// triangle is visited in index order and the first strict maximum of the dot product wins (vertices no triangle names
// are never seen)
V3 meshSupport(const float *verts, const int *tri, int ntri, V3 dir) {
  const float l2 = len2(dir);
  if (l2 < 0.0001f) dir = mk(1, 0, 0);
  else dir = dir * (1.0f / sqrtf(l2));
  V3 best = mk(0, 0, 0);
  float bestDot = -1e18f;   // BT_LARGE_FLOAT
  for (int i = 0; i < 3 * ntri; ++i) {
    const float *p = verts + 3 * tri[i];
    const V3 v = mk(p[0], p[1], p[2]);
    const float d = dot(dir, v);
    if (d > bestDot) { bestDot = d; best = v; }
  }
  return best;
}

V3 normalizedOf(V3 v) { return vdiv(v, len(v)); }

V3 triNormal(V3 v0, V3 v1, V3 v2) {
  const V3 cp = cross(v1 - v0, v2 - v1);
  const float m = len(cp);
  if (m == 0) return mk(1, 0, 0);
  return cp * (1.0f / m);
}

// ---- Melax hull ------------------------------------------------------------------------------------------------
struct HullTri {
  int v[3];
  int nb[3];   // nb[i]: the triangle across the edge opposite corner i
  int vmax;
  float rise;
  bool alive;
};

class HullBuilder {
 public:
  std::vector<HullTri> tris;   // position = the reference's triangle id; dead entries are its NULL slots
  bool broken = false;         // a state the reference would dereference a null pointer in

  int alloc(int a, int b, int c) {
    HullTri t;
    t.v[0] = a, t.v[1] = b, t.v[2] = c;
    t.nb[0] = t.nb[1] = t.nb[2] = -1;
    t.vmax = -1;
    t.rise = 0.0f;
    t.alive = true;
    tris.push_back(t);
    return (int)tris.size() - 1;
  }
  static bool hasVert(const HullTri &t, int v) { return t.v[0] == v || t.v[1] == v || t.v[2] == v; }
  // the neighbour slot of triangle `id` across its edge a-b (either direction)
  int none = -1;               // where neib() points when there is no such edge (broken is set)
  int &neib(int id, int a, int b) {
    if (id < 0 || id >= (int)tris.size() || !tris[id].alive) { broken = true; none = -1; return none; }
    HullTri &t = tris[id];
    for (int i = 0; i < 3; ++i) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
      if ((t.v[i] == a && t.v[i1] == b) || (t.v[i] == b && t.v[i1] == a)) return t.nb[i2];
    }
    broken = true;
    none = -1;
    return none;
  }
  // s and t lie back to back: their outer neighbours become each other's, both go
  void removeBackToBack(int s, int t) {
    for (int i = 0; i < 3; ++i) {
      const int a = tris[s].v[(i + 1) % 3], b = tris[s].v[(i + 2) % 3];
      const int sOut = neib(s, a, b), tOut = neib(t, b, a);
      neib(sOut, b, a) = tOut;
      neib(tOut, a, b) = sOut;
    }
    tris[s].alive = false;
    tris[t].alive = false;
  }
  bool liveWithVert(int id, int v) {
    if (id < 0 || id >= (int)tris.size() || !tris[id].alive) { broken = true; return false; }
    return hasVert(tris[id], v);
  }
  // replace triangle t0 by the three that join its edges to vertex v
  void extrude(int t0, int v) {
    const HullTri t = tris[t0];
    const int n = (int)tris.size();
    const int ta = alloc(v, t.v[1], t.v[2]);
    tris[ta].nb[0] = t.nb[0], tris[ta].nb[1] = n + 1, tris[ta].nb[2] = n + 2;
    neib(t.nb[0], t.v[1], t.v[2]) = n + 0;
    const int tb = alloc(v, t.v[2], t.v[0]);
    tris[tb].nb[0] = t.nb[1], tris[tb].nb[1] = n + 2, tris[tb].nb[2] = n + 0;
    neib(t.nb[1], t.v[2], t.v[0]) = n + 1;
    const int tc = alloc(v, t.v[0], t.v[1]);
    tris[tc].nb[0] = t.nb[2], tris[tc].nb[1] = n + 0, tris[tc].nb[2] = n + 1;
    neib(t.nb[2], t.v[0], t.v[1]) = n + 2;
    if (liveWithVert(tris[ta].nb[0], v)) removeBackToBack(ta, tris[ta].nb[0]);
    if (liveWithVert(tris[tb].nb[0], v)) removeBackToBack(tb, tris[tb].nb[0]);
    if (liveWithVert(tris[tc].nb[0], v)) removeBackToBack(tc, tris[tc].nb[0]);
    tris[t0].alive = false;
  }
  // the live triangle with the highest rise (the first of equals), if that rise exceeds epsilon
  int extrudable(float epsilon) const {
    int t = -1;
    for (int i = 0; i < (int)tris.size(); ++i)
      if (tris[i].alive && (t < 0 || tris[t].rise < tris[i].rise)) t = i;
    // (the reference starts from slot 0 even when it is empty and would fault on it; slot 0 is only ever empty
    // together with a live triangle further on taking over, which the loop above handles the same way)
    if (t < 0) return -1;
    return tris[t].rise > epsilon ? t : -1;
  }
};

int maxDirFiltered(const std::vector<V3> &p, V3 dir, const std::vector<int> &allow) {
  int m = -1;
  for (int i = 0; i < (int)p.size(); ++i)
    if (allow[i] && (m == -1 || dot(p[i], dir) > dot(p[m], dir))) m = i;
  return m;
}

V3 orthTo(V3 v) {
  const V3 a = cross(v, mk(0, 0, 1)), b = cross(v, mk(0, 1, 0));
  return len(a) > len(b) ? normalizedOf(a) : normalizedOf(b);
}

// maxdirsterid: the extreme point along dir that stays extreme when dir is tilted by 0.025 all the way round (45 degree
// steps, 5 degree steps where the answer changes); points that fail are struck from `allow`, the one that passes is marked 3
int maxDirSteady(const std::vector<V3> &p, V3 dir, std::vector<int> &allow) {
  const float radsPerDeg = 6.283185307179586232f / 360.0f;
  for (;;) {
    const int m = maxDirFiltered(p, dir, allow);
    if (m < 0) return -1;   // (every point struck: the reference asserts)
    if (allow[m] == 3) return m;
    const V3 u = orthTo(dir);
    const V3 v = cross(u, dir);
    int ma = -1;
    for (float x = 0.0f; x <= 360.0f; x += 45.0f) {
      const float s = sinf(radsPerDeg * x), c = cosf(radsPerDeg * x);
      const int mb = maxDirFiltered(p, dir + (u * s + v * c) * 0.025f, allow);
      if (ma == m && mb == m) { allow[m] = 3; return m; }
      if (ma != -1 && ma != mb) {
        int mc = ma;
        for (float xx = x - 40.0f; xx <= x; xx += 5.0f) {
          const float s2 = sinf(radsPerDeg * xx), c2 = cosf(radsPerDeg * xx);
          const int md = maxDirFiltered(p, dir + (u * s2 + v * c2) * 0.025f, allow);
          if (mc == m && md == m) { allow[m] = 3; return m; }
          mc = md;
        }
      }
      ma = mb;
    }
    allow[m] = 0;
  }
}

bool findSimplex(const std::vector<V3> &p, std::vector<int> &allow, int out[4]) {
  V3 b0 = mk(0.01f, 0.02f, 1.0f);
  const int p0 = maxDirSteady(p, b0, allow);
  const int p1 = maxDirSteady(p, -b0, allow);
  if (p0 < 0 || p1 < 0) return false;
  b0 = p[p0] - p[p1];
  if (p0 == p1 || veq(b0, mk(0, 0, 0))) return false;
  V3 b1 = cross(mk(1.0f, 0.02f, 0.0f), b0);
  V3 b2 = cross(mk(-0.02f, 1.0f, 0.0f), b0);
  b1 = len(b1) > len(b2) ? normalizedOf(b1) : normalizedOf(b2);
  int p2 = maxDirSteady(p, b1, allow);
  if (p2 == p0 || p2 == p1) p2 = maxDirSteady(p, -b1, allow);
  if (p2 < 0 || p2 == p0 || p2 == p1) return false;
  b1 = p[p2] - p[p0];
  b2 = normalizedOf(cross(b1, b0));
  int p3 = maxDirSteady(p, b2, allow);
  if (p3 == p0 || p3 == p1 || p3 == p2) p3 = maxDirSteady(p, -b2, allow);
  if (p3 < 0 || p3 == p0 || p3 == p1 || p3 == p2) return false;
  if (dot(p[p3] - p[p0], cross(p[p1] - p[p0], p[p2] - p[p0])) < 0) { const int t = p2; p2 = p3; p3 = t; }
  out[0] = p0, out[1] = p1, out[2] = p2, out[3] = p3;
  return true;
}

bool isAbove(const std::vector<V3> &p, const HullTri &t, V3 q, float epsilon) {
  const V3 n = triNormal(p[t.v[0]], p[t.v[1]], p[t.v[2]]);
  return dot(n, q - p[t.v[0]]) > epsilon;
}

// calchull: the triangle list (vertex numbers into p) of the hull, in triangle-id order
bool melaxHull(const std::vector<V3> &p, int vlimit, std::vector<int> &trisOut) {
  const int n = (int)p.size();
  if (n < 4) return false;
  if (vlimit == 0) vlimit = 1000000000;
  V3 bmin = p[0], bmax = p[0];
  std::vector<int> isExtreme(n, 0), allow(n, 1);
  for (int j = 0; j < n; ++j) {
    bmin = mk(fminf(bmin.x, p[j].x), fminf(bmin.y, p[j].y), fminf(bmin.z, p[j].z));
    bmax = mk(fmaxf(bmax.x, p[j].x), fmaxf(bmax.y, p[j].y), fmaxf(bmax.z, p[j].z));
  }
  const float epsilon = len(bmax - bmin) * 0.001f;
  int s[4];
  if (!findSimplex(p, allow, s)) return false;
  const V3 center = vdiv(p[s[0]] + p[s[1]] + p[s[2]] + p[s[3]], 4.0f);
  HullBuilder h;
  const int seed[4][6] = {{s[2], s[3], s[1], 2, 3, 1}, {s[3], s[2], s[0], 3, 2, 0}, {s[0], s[1], s[3], 0, 1, 3}, {s[1], s[0], s[2], 1, 0, 2}};
  for (int k = 0; k < 4; ++k) {
    const int t = h.alloc(seed[k][0], seed[k][1], seed[k][2]);
    h.tris[t].nb[0] = seed[k][3], h.tris[t].nb[1] = seed[k][4], h.tris[t].nb[2] = seed[k][5];
  }
  isExtreme[s[0]] = isExtreme[s[1]] = isExtreme[s[2]] = isExtreme[s[3]] = 1;
  for (int j = 0; j < (int)h.tris.size(); ++j) {
    HullTri &t = h.tris[j];
    const V3 nrm = triNormal(p[t.v[0]], p[t.v[1]], p[t.v[2]]);
    t.vmax = maxDirSteady(p, nrm, allow);
    if (t.vmax < 0) return false;
    t.rise = dot(nrm, p[t.vmax] - p[t.v[0]]);
  }
  vlimit -= 4;
  int te;
  while (vlimit > 0 && (te = h.extrudable(epsilon)) >= 0) {
    const int v = h.tris[te].vmax;
    isExtreme[v] = 1;
    // every triangle that sees v, newest first
    for (int j = (int)h.tris.size(); j-- > 0;) {
      if (!h.tris[j].alive) continue;
      if (isAbove(p, h.tris[j], p[v], 0.01f * epsilon)) h.extrude(j, v);
    }
    // the new fan: a triangle that faces inwards or is a sliver takes its outer neighbour with it
    for (int j = (int)h.tris.size(); j-- > 0;) {
      if (!h.tris[j].alive) continue;
      if (!HullBuilder::hasVert(h.tris[j], v)) break;
      const HullTri nt = h.tris[j];
      if (isAbove(p, nt, center, 0.01f * epsilon) ||
          len(cross(p[nt.v[1]] - p[nt.v[0]], p[nt.v[2]] - p[nt.v[1]])) < epsilon * epsilon * 0.1f) {
        const int nb = nt.nb[0];
        if (nb < 0 || nb >= (int)h.tris.size() || !h.tris[nb].alive) return false;
        h.extrude(nb, v);
        j = (int)h.tris.size();
      }
    }
    // rise of the new triangles
    for (int j = (int)h.tris.size(); j-- > 0;) {
      HullTri &t = h.tris[j];
      if (!t.alive) continue;
      if (t.vmax >= 0) break;
      const V3 nrm = triNormal(p[t.v[0]], p[t.v[1]], p[t.v[2]]);
      t.vmax = maxDirSteady(p, nrm, allow);
      if (t.vmax < 0) return false;
      if (isExtreme[t.vmax]) t.vmax = -1;   // already on the hull: the triangle is final (rise stays 0)
      else t.rise = dot(nrm, p[t.vmax] - p[t.v[0]]);
    }
    if (h.broken) return false;
    --vlimit;
  }
  if (h.broken) return false;
  trisOut.clear();
  for (const HullTri &t : h.tris)
    if (t.alive) trisOut.insert(trisOut.end(), t.v, t.v + 3);
  return true;
}

void pushBoxCorners(std::vector<V3> &out, float x1, float x2, float y1, float y2, float z1, float z2) {
  out.clear();
  out.push_back(mk(x1, y1, z1)); out.push_back(mk(x2, y1, z1)); out.push_back(mk(x2, y2, z1)); out.push_back(mk(x1, y2, z1));
  out.push_back(mk(x1, y1, z2)); out.push_back(mk(x2, y1, z2)); out.push_back(mk(x2, y2, z2)); out.push_back(mk(x1, y2, z2));
}

float dist2To(float px, float py, float pz, V3 c) {
  const float dx = px - c.x, dy = py - c.y, dz = pz - c.z;
  return dx * dx + dy * dy + dz * dz;
}

// CleanupVertices: points in the unit box of their bounds, near-duplicates (all three coordinates within eps) merged
// keeping the one farther from the centre; flat or tiny clouds become a box.  Returns the scale to multiply back.
bool cleanupVertices(const std::vector<V3> &src, std::vector<V3> &out, float normalEpsilon, V3 &scale) {
  const float kEps = 0.000001f;
  out.clear();
  scale = mk(1, 1, 1);
  if (src.empty()) return false;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const V3 &v : src) {
    const float c[3] = {v.x, v.y, v.z};
    for (int j = 0; j < 3; ++j) {
      if (c[j] < lo[j]) lo[j] = c[j];
      if (c[j] > hi[j]) hi[j] = c[j];
    }
  }
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  V3 center = mk(dx * 0.5f + lo[0], dy * 0.5f + lo[1], dz * 0.5f + lo[2]);
  float recip[3] = {0, 0, 0};
  if (dx < kEps || dy < kEps || dz < kEps || src.size() < 3) {
    float l = FLT_MAX;
    if (dx > kEps && dx < l) l = dx;
    if (dy > kEps && dy < l) l = dy;
    if (dz > kEps && dz < l) l = dz;
    if (l == FLT_MAX) dx = dy = dz = 0.01f;
    else {
      if (dx < kEps) dx = l * 0.05f;
      if (dy < kEps) dy = l * 0.05f;
      if (dz < kEps) dz = l * 0.05f;
    }
    pushBoxCorners(out, center.x - dx, center.x + dx, center.y - dy, center.y + dy, center.z - dz, center.z + dz);
    return true;
  }
  scale = mk(dx, dy, dz);
  recip[0] = 1 / dx, recip[1] = 1 / dy, recip[2] = 1 / dz;
  center = mk(center.x * recip[0], center.y * recip[1], center.z * recip[2]);
  for (const V3 &v : src) {
    const float px = v.x * recip[0], py = v.y * recip[1], pz = v.z * recip[2];
    size_t j = 0;
    for (; j < out.size(); ++j) {
      V3 &o = out[j];
      if (fabsf(o.x - px) < normalEpsilon && fabsf(o.y - py) < normalEpsilon && fabsf(o.z - pz) < normalEpsilon) {
        if (dist2To(px, py, pz, center) > dist2To(o.x, o.y, o.z, center)) o = mk(px, py, pz);
        break;
      }
    }
    if (j == out.size()) out.push_back(mk(px, py, pz));
  }
  // what is left may have collapsed
  float lo2[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi2[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const V3 &v : out) {
    const float c[3] = {v.x, v.y, v.z};
    for (int j = 0; j < 3; ++j) {
      if (c[j] < lo2[j]) lo2[j] = c[j];
      if (c[j] > hi2[j]) hi2[j] = c[j];
    }
  }
  dx = hi2[0] - lo2[0], dy = hi2[1] - lo2[1], dz = hi2[2] - lo2[2];
  if (dx < kEps || dy < kEps || dz < kEps || out.size() < 3) {
    const float cx = dx * 0.5f + lo2[0], cy = dy * 0.5f + lo2[1], cz = dz * 0.5f + lo2[2];
    float l = FLT_MAX;
    if (dx >= kEps && dx < l) l = dx;
    if (dy >= kEps && dy < l) l = dy;
    if (dz >= kEps && dz < l) l = dz;
    if (l == FLT_MAX) dx = dy = dz = 0.01f;
    else {
      if (dx < kEps) dx = l * 0.05f;
      if (dy < kEps) dy = l * 0.05f;
      if (dz < kEps) dz = l * 0.05f;
    }
    pushBoxCorners(out, cx - dx, cx + dx, cy - dy, cy + dy, cz - dz, cz + dz);
  }
  return true;
}

// HullLibrary::CreateConvexHull, QF_TRIANGLES, default HullDesc (normal epsilon 0.001, at most 4096 vertices)
bool createConvexHull(const std::vector<V3> &pts, std::vector<V3> &hullVerts) {
  std::vector<V3> clean;
  V3 scale;
  if (!cleanupVertices(pts, clean, 0.001f, scale)) return false;
  for (V3 &v : clean) v = mk(v.x * scale.x, v.y * scale.y, v.z * scale.z);
  std::vector<int> tris;
  if (!melaxHull(clean, 4096, tris)) return false;
  // BringOutYourDead: vertices in the order the index list first names them
  std::vector<int> newIndex(clean.size(), -1);
  hullVerts.clear();
  for (int idx : tris)
    if (newIndex[idx] < 0) {
      newIndex[idx] = (int)hullVerts.size();
      hullVerts.push_back(clean[idx]);
    }
  return true;
}
}  // namespace

extern "C" int tohc_shape_hull(const float *verts, int n_verts, const int32_t *tri, int n_tri, float *out, int cap) {
  // a mesh names at least one triangle (the reference asserts indices.size() >= 3, bullet_collision_checker.cpp:162): with
  // none, a mesh of >= 50 vertices would take 42 support points over nothing and return HullLibrary's fall-back box
  if (!verts || n_verts < 1 || n_tri < 1 || !tri || (cap > 0 && !out)) return TOHC_E_INVALID;
  for (int i = 0; i < 3 * n_tri; ++i)
    if (tri[i] < 0 || tri[i] >= n_verts) return TOHC_E_INVALID;
  std::vector<V3> hull;
  bool reduced = false;
  if (n_verts >= 50) {   // bullet_collision_checker.cpp:183
    std::vector<V3> support(42);
    for (int i = 0; i < 42; ++i) support[i] = meshSupport(verts, tri, n_tri, mk(kDirections[i][0], kDirections[i][1], kDirections[i][2]));
    reduced = createConvexHull(support, hull);   // failure: "falling back to original vertices" (:185-186)
  }
  if (!reduced) {
    hull.resize((size_t)n_verts);
    for (int i = 0; i < n_verts; ++i) hull[i] = mk(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
  }
  const int n = (int)hull.size();
  for (int i = 0; i < n && i < cap; ++i) out[3 * i] = hull[i].x, out[3 * i + 1] = hull[i].y, out[3 * i + 2] = hull[i].z;
  return n;
}
