// emu.cpp -- HOST EMULATION of the device narrowphase (trajopt_amd/csrc/tohc_narrow.h compiled for the CPU with a
// group size of one lane).  TEST-ONLY: it lets the CPU test-suite check the device algorithm restatement against the
// oracle without a GPU.  The product never loads this library; the shipped path is the HIP build of the same header.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define TOHC_EMU_STATS 1
#include "../../oracle/oracle_api.h"
#include "../../trajopt_amd/csrc/tohc_narrow.h"

using namespace tohc;
namespace {
struct Shape { int kind; V3 dims, lmin, lmax; int voff, n; };
struct World {
  std::vector<Shape> shapes;
  std::vector<float> vx, vy, vz;
  std::vector<F4> v4;
  std::vector<float> v8;
  std::vector<int> voff8;
  std::vector<int> child_shape;
  std::vector<Xf> child_T;
  std::vector<int> off, cnt, compound, group;
  std::vector<unsigned char> allowed;
  float d, breaking;
  long pair_evals, epa_evals;
};
float safeMargin(float cur, float x, float y, float z) {
  float mn = x < y ? (x < z ? x : z) : (y < z ? y : z);
  float safe = 0.1f * mn;
  return safe < cur ? safe : cur;
}
void load(const World *w, int child, Cvx<1> &c) {
  const Shape &s = w->shapes[w->child_shape[child]];
  cvxLoad<1>(c, s.kind, s.dims, w->vx.data(), w->vy.data(), w->vz.data(), w->v4.data(), s.voff, s.n, w->v8.data(), w->voff8[w->child_shape[child]]);
}
Xf childWorld(const World *w, int obj, int c, const Xf &T) { return w->compound[obj] ? xfMul(T, w->child_T[w->off[obj] + c]) : T; }
long g_iter_hist[64];
long g_epa_hist[2];
long g_epa_redone;
long g_pre[8];
int g_log_q = -1, g_log_x = -1;   // (EMU_EPA_LOG: scene-wide child numbers of the pair being evaluated)
bool pairEval(World *w, const Cvx<1> &A, const Xf &tA, const Cvx<1> &B, const Xf &tB, ora_contact *c) {
  const float md = pairMaxDist2<1>(A, B, w->breaking);
  GjkState st;
  w->pair_evals++;
  if (isSphereSphere<1>(A, B)) {   // as k_gjk does
    PairOut o;
    sphereSphere(tA.o, shapeMargin<1>(A), tB.o, shapeMargin<1>(B), o);
    if (!o.valid || (double)o.distance > (double)w->d) return false;
    V3 pA = o.pointOnB + o.normalOnB * o.distance;
    c->dist = o.distance;
    c->n[0] = o.normalOnB.x; c->n[1] = o.normalOnB.y; c->n[2] = o.normalOnB.z;
    c->ptA[0] = pA.x; c->ptA[1] = pA.y; c->ptA[2] = pA.z;
    c->ptB[0] = o.pointOnB.x; c->ptB[1] = o.pointOnB.y; c->ptB[2] = o.pointOnB.z;
    return true;
  }
  int pre = 0;
  {
    float lim = w->d * 1.01f + 1e-4f;
    V3 ww;
    V3 dir = tA.o - tB.o;
    if (len2(dir) > 1e-12f && sepAxisBeyond<1>(A, B, tA, tB, dir, lim, ww)) pre = 1;
    else if (len2(ww) > 1e-12f && sepAxisBeyond<1>(A, B, tA, tB, ww, lim, ww)) pre = 2;
    else if (len2(ww) > 1e-12f && sepAxisBeyond<1>(A, B, tA, tB, ww, lim, ww)) pre = 3;
  }
  g_pre[pre]++;
  gjkMain<1>(A, B, tA, tB, md, st);
  const int gjkIters = st.iters;   // (EMU_EPA_LOG)
  if (const char *lg = getenv("EMU_GJK_LOG")) {   // diagnostics: what a GJK run's length depends on (pairs the pre-test passes on)
    static FILE *f = fopen(lg, "w");
    V3 dir = tA.o - tB.o, ww;
    float delta = 0;
    if (len2(dir) > 1e-12f) { dir = vdiv(dir, len(dir)); sepAxisBeyond<1>(A, B, tA, tB, dir, 0.f, ww); delta = dot(dir, ww); }
    if (!pre) fprintf(f, "%d %g %d %d %d\n", st.iters, (double)delta, A.n, B.n, (int)st.needEpa), fflush(f);
  }
  g_iter_hist[st.iters < 63 ? st.iters : 63]++;
  g_epa_hist[st.needEpa ? 1 : 0]++;
  if (st.needEpa) {
    // the two stages exactly as the device runs them: k_pen on a G2Mem, hand-over, k_epa on an EpaMem
    static EpaMem mem;
    static G2Mem mem2;
    w->epa_evals++;
    Xf lA, lB;
    penLocal(tA, tB, st.offset, lA, lB);
    V3 tmpA = mk(0, 0, 0), tmpB = mk(0, 0, 0), sep = mk(0, 0, 0);
    bool isValid2 = false;
    Gjk2<1, G2Mem> g1;
    if (penStage1<1, G2Mem>(&mem2, A, B, lA, lB, g1, isValid2, sep, tmpA, tmpB)) {
      PenHand h;
      penSave<1, G2Mem>(h, g1);
      // k_tail_epa: first on the small pools, again on the reference's full ones when the run would leave them
      static EpaMemS memS;
      Gjk2<1, EpaMemS> gs;
      penRestore<1>(&memS, h, A, B, lA, lB, gs);
      if (!penStage2<1>(&memS, lA, lB, gs, isValid2, sep, tmpA, tmpB)) {
        g_epa_redone++;
        Gjk2<1, EpaMem> g2;
        penRestore<1>(&mem, h, A, B, lA, lB, g2);
        penStage2<1>(&mem, lA, lB, g2, isValid2, sep, tmpA, tmpB);
      }
    }
    gjkFinishTail(isValid2, sep, tmpA, tmpB, st);
    if (const char *lg = getenv("EMU_EPA_LOG")) {   // diagnostics: what an EPA run's length depends on
      static FILE *f = fopen(lg, "w");
      static long prevIt = 0, prevCalls = 0;
      const EmuStats &es = emuStats();
      V3 dir = tA.o - tB.o, ww;
      float ov = 0;
      if (len2(dir) > 1e-12f) { dir = vdiv(dir, len(dir)); sepAxisBeyond<1>(A, B, tA, tB, dir, 0.f, ww); ov = -dot(dir, ww); }
      if (es.epa_calls != prevCalls) fprintf(f, "%ld %d %d %d %d %g %g %d %d %ld %g %d\n", es.epa_iters - prevIt, A.kind, A.n, B.kind, B.n, (double)st.distance, (double)ov, g_log_q, g_log_x, es.last_g2, es.last_d0, gjkIters), fflush(f);
      prevIt = es.epa_iters; prevCalls = es.epa_calls;
    }
  }
  PairOut o;
  gjkOutput(st, md, o);
  if (!o.valid) return false;
  if ((double)o.distance > (double)w->d) return false;
  if (pre) g_pre[4 + pre]++;
  V3 pA = o.pointOnB + o.normalOnB * o.distance;
  c->dist = o.distance;
  c->n[0] = o.normalOnB.x; c->n[1] = o.normalOnB.y; c->n[2] = o.normalOnB.z;
  c->ptA[0] = pA.x; c->ptA[1] = pA.y; c->ptA[2] = pA.z;
  c->ptB[0] = o.pointOnB.x; c->ptB[1] = o.pointOnB.y; c->ptB[2] = o.pointOnB.z;
  return true;
}
}  // namespace

extern "C" {
void *emu_create(const ora_scene *sc) {
  World *w = new World;
  for (int i = 0; i < sc->n_shapes; ++i) {
    Shape s;
    s.kind = sc->shape_kind[i];
    const float *p = sc->shape_param + 4 * i;
    s.voff = sc->shape_voff[i];
    s.n = sc->shape_vcnt[i];
    if (s.kind == ORA_BOX) {
      float m = safeMargin(0.04f, p[0], p[1], p[2]);
      s.dims = mk(((p[0] * 1.0f - m) + m) - 0.0f, ((p[1] * 1.0f - m) + m) - 0.0f, ((p[2] * 1.0f - m) + m) - 0.0f);
      s.lmin = -s.dims; s.lmax = s.dims;
    } else if (s.kind == ORA_SPHERE) {
      s.dims = mk(p[0], p[0], p[0]);
      s.lmin = -s.dims; s.lmax = s.dims;
    } else if (s.kind == ORA_CYLZ) {
      float m = safeMargin(0.04f, p[0], p[0], p[1]);
      s.dims = mk(((p[0] * 1.0f - m) + m) - 0.0f, ((p[0] * 1.0f - m) + m) - 0.0f, ((p[1] * 1.0f - m) + m) - 0.0f);
      s.lmin = -s.dims; s.lmax = s.dims;
    } else {
      s.dims = mk(0, 0, 0);
      s.lmin = mk(INFINITY, INFINITY, INFINITY); s.lmax = mk(-INFINITY, -INFINITY, -INFINITY);
      for (int k = 0; k < s.n; ++k) {
        const float *q = sc->verts + 3 * (s.voff + k);
        s.lmin = mk(fminf(s.lmin.x, q[0]), fminf(s.lmin.y, q[1]), fminf(s.lmin.z, q[2]));
        s.lmax = mk(fmaxf(s.lmax.x, q[0]), fmaxf(s.lmax.y, q[1]), fmaxf(s.lmax.z, q[2]));
      }
    }
    w->shapes.push_back(s);
  }
  for (int i = 0; i < sc->n_verts; ++i) { w->vx.push_back(sc->verts[3 * i]); w->vy.push_back(sc->verts[3 * i + 1]); w->vz.push_back(sc->verts[3 * i + 2]); }
  for (size_t i = 0; i < w->vx.size(); ++i) { F4 p; p.x = w->vx[i]; p.y = w->vy[i]; p.z = w->vz[i]; p.w = 0; w->v4.push_back(p); }
  { F4 z; z.x = z.y = z.z = z.w = 0; w->v4.resize(w->v4.size() + 64, z); }
  w->vx.resize(w->vx.size() + 64); w->vy.resize(w->vy.size() + 64); w->vz.resize(w->vz.size() + 64);
  {
    std::vector<int> kind, voff, vcnt;
    for (const Shape &s : w->shapes) { kind.push_back(s.kind); voff.push_back(s.voff); vcnt.push_back(s.n); }
    blockVertices(w->vx.data(), w->vy.data(), w->vz.data(), (int)w->shapes.size(), kind.data(), voff.data(), vcnt.data(), w->v8, w->voff8);
  }
  for (int i = 0; i < sc->n_children; ++i) { w->child_shape.push_back(sc->child_shape[i]); w->child_T.push_back(xfFrom7(sc->child_T + 7 * i)); }
  for (int i = 0; i < sc->n_objects; ++i) {
    w->off.push_back(sc->obj_child_off[i]); w->cnt.push_back(sc->obj_child_cnt[i]);
    w->compound.push_back(sc->obj_compound[i]); w->group.push_back(sc->obj_group[i]);
  }
  w->allowed.assign((size_t)sc->n_objects * sc->n_objects, 1);
  w->d = 0.05f; w->breaking = (float)(2.001 * (double)w->d);
  w->pair_evals = w->epa_evals = 0;
  return w;
}
void emu_destroy(void *h) { delete (World *)h; }
void emu_set_contact_distance(void *h, float d) { World *w = (World *)h; w->d = d; w->breaking = (float)(2.001 * (double)d); }
void emu_set_allowed(void *h, const unsigned char *m) { World *w = (World *)h; memcpy(w->allowed.data(), m, w->allowed.size()); }

int emu_links_vs_all(void *h, const double *objT7, const int *q, int nq, int filterMask, ora_contact *out, int cap) {
  World *w = (World *)h;
  int n = 0, no = (int)w->off.size();
  std::vector<Xf> T(no);
  for (int i = 0; i < no; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    for (int X = 0; X < no; ++X) {
      if (X == Q) continue;
      if (!(w->group[X] & (short)filterMask)) continue;
      if (!w->allowed[(size_t)Q * no + X]) continue;
      for (int qc = 0; qc < w->cnt[Q]; ++qc) {
        Xf tq = childWorld(w, Q, qc, T[Q]);
        Cvx<1> cq; load(w, w->off[Q] + qc, cq);
        const Shape &sq = w->shapes[w->child_shape[w->off[Q] + qc]];
        Aabb bq = shapeAabb(sq.lmin, sq.lmax, tq, w->d);
        for (int xc = 0; xc < w->cnt[X]; ++xc) {
          Xf tx = childWorld(w, X, xc, T[X]);
          Cvx<1> cx; load(w, w->off[X] + xc, cx);
          const Shape &sx = w->shapes[w->child_shape[w->off[X] + xc]];
          if (!aabbOverlap(bq, shapeAabb(sx.lmin, sx.lmax, tx, w->d))) continue;
          ora_contact c;
          memset(&c, 0, sizeof c);
          bool kept;
          g_log_q = w->off[Q] + qc; g_log_x = w->off[X] + xc;
          if (w->compound[X]) { kept = pairEval(w, cx, tx, cq, tq, &c); c.objA = X; c.childA = xc; c.objB = Q; c.childB = qc; }
          else { kept = pairEval(w, cq, tq, cx, tx, &c); c.objA = Q; c.childA = qc; c.objB = X; c.childB = xc; }
          if (kept) { if (n < cap) out[n] = c; ++n; }
        }
      }
    }
  }
  return n;
}

int emu_cast_vs_all(void *h, const double *objT7, const int *q, int nq, const double *T0_7, const double *T1_7, ora_contact *out, int cap) {
  World *w = (World *)h;
  int n = 0, no = (int)w->off.size();
  std::vector<Xf> T(no);
  for (int i = 0; i < no; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    Xf L0 = xfFrom7(T0_7 + 7 * qi), L1 = xfFrom7(T1_7 + 7 * qi);
    for (int qc = 0; qc < w->cnt[Q]; ++qc) {
      Xf tf0 = L0, tf1 = L1;
      if (w->compound[Q]) { tf0 = xfMul(L0, w->child_T[w->off[Q] + qc]); tf1 = xfMul(L1, w->child_T[w->off[Q] + qc]); }
      Cvx<1> cq; load(w, w->off[Q] + qc, cq);
      cq.cast = 1;
      cq.t01 = xfInverseTimes(tf0, tf1);
      Xf tw1 = xfMul(tf0, cq.t01);
      const Shape &sq = w->shapes[w->child_shape[w->off[Q] + qc]];
      Aabb bq = aabbUnion(shapeAabb(sq.lmin, sq.lmax, tf0, w->d), shapeAabb(sq.lmin, sq.lmax, tw1, w->d));
      for (int X = 0; X < no; ++X) {
        if (!(w->group[X] & ORA_GROUP_KINBODY)) continue;
        if (!w->allowed[(size_t)Q * no + X]) continue;
        for (int xc = 0; xc < w->cnt[X]; ++xc) {
          Xf tx = childWorld(w, X, xc, T[X]);
          Cvx<1> cx; load(w, w->off[X] + xc, cx);
          const Shape &sx = w->shapes[w->child_shape[w->off[X] + xc]];
          if (!aabbOverlap(bq, shapeAabb(sx.lmin, sx.lmax, tx, w->d))) continue;
          ora_contact c;
          memset(&c, 0, sizeof c);
          bool kept, castFirst;
          if (w->compound[X] && cq.kind == KIND_SPHERE && cx.kind == KIND_SPHERE && castSphereChildCulled(tf0.o, xfApply(tf0, cq.t01.o), shapeMargin<1>(cq), tx.o, shapeMargin<1>(cx))) {
            w->pair_evals++;   // as k_gjk does: the evaluation ends at the reference's child-box test
            continue;
          }
          g_log_q = w->off[Q] + qc; g_log_x = w->off[X] + xc;
          if (w->compound[X]) { kept = pairEval(w, cx, tx, cq, tf0, &c); c.objA = X; c.childA = xc; c.objB = Q; c.childB = qc; castFirst = false; }
          else { kept = pairEval(w, cq, tf0, cx, tx, &c); c.objA = Q; c.childA = qc; c.objB = X; c.childB = xc; castFirst = true; }
          if (!kept) continue;
          V3 nW = mk((float)c.n[0], (float)c.n[1], (float)c.n[2]);
          V3 pA = mk((float)c.ptA[0], (float)c.ptA[1], (float)c.ptA[2]), pB = mk((float)c.ptB[0], (float)c.ptB[1], (float)c.ptB[2]);
          CastOut co;
          castClassify<1>(cq, tf0, tw1, nW, pA, pB, castFirst, co);
          if (castFirst) {
            for (int k = 0; k < 3; ++k) { double t = c.ptA[k]; c.ptA[k] = c.ptB[k]; c.ptB[k] = t; c.n[k] *= -1; }
            int t = c.objA; c.objA = c.objB; c.objB = t;
            t = c.childA; c.childA = c.childB; c.childB = t;
          }
          c.time = co.time; c.cctype = co.cctype;
          if (co.between) {
            c.ptB[0] = co.ptB.x; c.ptB[1] = co.ptB.y; c.ptB[2] = co.ptB.z;
            c.ptB1[0] = co.ptB1.x; c.ptB1[1] = co.ptB1.y; c.ptB1[2] = co.ptB1.z;
          }
          if (n < cap) out[n] = c;
          ++n;
        }
      }
    }
  }
  return n;
}
// ContinuousCheckTrajectory with the device's timeOfImpact<1> (same loops and tie rule as ora_continuous_check)
int emu_continuous_check(void *h, const double *objT7, const int *q, int nq, const double *traj7, int nsteps, ora_sweep_hit *out, int cap) {
  World *w = (World *)h;
  static EpaMem mem;
  int n = 0, no = (int)w->off.size();
  std::vector<Xf> T(no);
  for (int i = 0; i < no; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int qi = 0; qi < nq; ++qi) {
    int Q = q[qi];
    for (int qc = 0; qc < w->cnt[Q]; ++qc) {
      Cvx<1> cq; load(w, w->off[Q] + qc, cq);
      const Shape &sq = w->shapes[w->child_shape[w->off[Q] + qc]];
      const float discA = angularMotionDisc(sq.kind, sq.dims, sq.lmin, sq.lmax, w->d);
      for (int seg = 0; seg + 1 < nsteps; ++seg) {
        Xf t0 = xfFrom7(traj7 + 7 * ((size_t)seg * nq + qi)), t1 = xfFrom7(traj7 + 7 * ((size_t)(seg + 1) * nq + qi));
        if (w->compound[Q]) { t0 = xfMul(t0, w->child_T[w->off[Q] + qc]); t1 = xfMul(t1, w->child_T[w->off[Q] + qc]); }
        float closest = 1.0f;
        bool have = false;
        ora_sweep_hit hit;
        memset(&hit, 0, sizeof hit);
        for (int X = 0; X < no; ++X) {
          if (!(w->group[X] & ORA_GROUP_KINBODY)) continue;
          for (int xc = 0; xc < w->cnt[X]; ++xc) {
            if (closest == 0.0f) break;
            Xf tx = childWorld(w, X, xc, T[X]);
            Cvx<1> cx; load(w, w->off[X] + xc, cx);
            const Shape &sx = w->shapes[w->child_shape[w->off[X] + xc]];
            SweepOut so;
            if (!timeOfImpact<1>(&mem, cq, discA, t0, t1, cx, angularMotionDisc(sx.kind, sx.dims, sx.lmin, sx.lmax, w->d), tx, so)) continue;
            if (!(so.fraction < closest)) continue;
            closest = so.fraction;
            have = true;
            hit.obj = Q; hit.child = qc; hit.seg = seg; hit.hitObj = X; hit.hitChild = xc; hit.fraction = so.fraction;
            hit.n[0] = so.normal.x; hit.n[1] = so.normal.y; hit.n[2] = so.normal.z;
            hit.pt[0] = so.point.x; hit.pt[1] = so.point.y; hit.pt[2] = so.point.z;
          }
        }
        if (have) { if (n < cap) out[n] = hit; ++n; }
      }
    }
  }
  return n;
}
float emu_pt_trig(int which, float x) { return which == 0 ? ptSin(x) : which == 1 ? ptCos(x) : ptAcos(x); }
int emu_ray_cast(void *h, const double *objT7, const double *from3, const double *to3, int n_rays, float *closest) {
  World *w = (World *)h;
  int hits = 0, no = (int)w->off.size();
  std::vector<Xf> T(no);
  for (int i = 0; i < no; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  for (int i = 0; i < n_rays; ++i) {
    const V3 from = mk((float)from3[3 * i], (float)from3[3 * i + 1], (float)from3[3 * i + 2]);
    const V3 to = mk((float)to3[3 * i], (float)to3[3 * i + 1], (float)to3[3 * i + 2]);
    float cl = 1.0f;
    for (int X = 0; X < no; ++X)
      for (int xc = 0; xc < w->cnt[X]; ++xc) {
        Xf tx = childWorld(w, X, xc, T[X]);
        Cvx<1> cx; load(w, w->off[X] + xc, cx);
        float fr;
        V3 nrm;
        if (rayConvex(cx, tx, from, to, fr, nrm) && fr < cl) cl = fr;   // the kernel: atomicMin over all children
      }
    closest[i] = cl;
    hits += cl < 1.0f;
  }
  return hits;
}
// Experiment: could k_ray cull a cast when the segment stays clear of a sphere that holds the shape, grown
// by 1.25 cm (the cast's loop ends when the closest vector is <= 1 cm long)?  No: btSubsimplexConvexCast keeps simplex
// vertices computed at earlier lambdas, so it sometimes reports a hit at a point that is farther from the shape than its
// tolerance (6 of 1.96e7 casts, up to 2 cm beyond the sphere).  out[0] casts, out[1] culled, out[2] culled casts that
// report a hit all the same, out[3] hits.
static bool rayMissesSphere(V3 from, V3 to, V3 c, float reach) {
  const V3 d = to - from, m = from - c;
  const float dd = dot(d, d);
  float t = dd > 0.0f ? -dot(m, d) / dd : 0.0f;
  t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
  const V3 p = m + d * t;
  return dot(p, p) > reach * reach;
}
void emu_ray_cull_probe(void *h, const double *objT7, const double *from3, const double *to3, int n_rays, long *out) {
  World *w = (World *)h;
  int no = (int)w->off.size();
  std::vector<Xf> T(no);
  for (int i = 0; i < no; ++i) T[i] = xfFrom7(objT7 + 7 * i);
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int X = 0; X < no; ++X)
    for (int xc = 0; xc < w->cnt[X]; ++xc) {
      Xf tx = childWorld(w, X, xc, T[X]);
      Cvx<1> cx; load(w, w->off[X] + xc, cx);
      const Shape &sh = w->shapes[w->child_shape[w->off[X] + xc]];
      const V3 c = xfApply(tx, (sh.lmin + sh.lmax) * 0.5f);
      const float reach = len((sh.lmax - sh.lmin) * 0.5f) * 1.0001f + 0.0125f;
      for (int i = 0; i < n_rays; ++i) {
        const V3 from = mk((float)from3[3 * i], (float)from3[3 * i + 1], (float)from3[3 * i + 2]);
        const V3 to = mk((float)to3[3 * i], (float)to3[3 * i + 1], (float)to3[3 * i + 2]);
        float fr;
        V3 nrm;
        const bool cull = rayMissesSphere(from, to, c, reach);
        const bool hit = rayConvex(cx, tx, from, to, fr, nrm) && fr < 1.0f;
        out[0]++; out[1] += cull; out[2] += cull && hit; out[3] += hit;
        if (cull && hit && getenv("EMU_CULL_DBG")) { const V3 d = to - from, m = from - c; float t = -dot(m, d) / dot(d, d); t = t < 0 ? 0 : (t > 1 ? 1 : t); fprintf(stderr, "CULLHIT ray %d obj %d child %d kind %d fr %.6f segdist %.4f reach %.4f\n", i, X, xc, cx.kind, fr, len(m + d * t), reach); }
      }
    }
}
void emu_set_trig_mode(int portable) { emuTrigMode() = portable; }
long emu_stat(int i) {
  const EmuStats &s = emuStats();
  const long *p = &s.gjk2_calls;
  return p[i];
}
long emu_pre(int i) { return g_pre[i]; }
long emu_iter_hist(int i) { return g_iter_hist[i]; }
long emu_epa_redone() { return g_epa_redone; }
long emu_counter(void *h, int which) { World *w = (World *)h; return which == 0 ? w->pair_evals : w->epa_evals; }
}
