// san_driver.cpp -- TEST INFRASTRUCTURE ("CPU restatement built with -fsanitize=address,undefined").
// A self-contained executable that drives the two CPU builds of the path's algorithm -- the C oracle (trajopt_oracle.c)
// and the host build of the DEVICE narrowphase source (tests/emu/emu.cpp over trajopt_amd/csrc/tohc_narrow.h) -- through
// random scenes (boxes, hulls, spheres, cylinders, compounds; discrete, cast, sweeps, rays, FK + linearization) with
// AddressSanitizer and UndefinedBehaviorSanitizer compiled in (oracle/Makefile target `san`).  It also cross-checks the
// two against each other bit for bit, so a run that prints "san_driver: OK" is a clean sanitizer run AND a parity run.
// A standalone binary on purpose: a sanitized shared library inside the Python test process would need the sanitizer
// runtime preloaded into the interpreter.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle_api.h"

extern "C" {
void *ora_create(const ora_scene *);
void ora_destroy(void *);
void ora_set_contact_distance(void *, float);
int ora_links_vs_all(void *, const double *, const int *, int, int, ora_contact *, int);
int ora_cast_vs_all(void *, const double *, const int *, int, const double *, const double *, ora_contact *, int);
int ora_continuous_check(void *, const double *, const int *, int, const double *, int, ora_sweep_hit *, int);
int ora_ray_cast(void *, const double *, const double *, const double *, int, float *);
int ora_robot_dof(const ora_robot *);
void ora_fk(const ora_robot *, const double *, double *);
int ora_linearize_discrete(void *, const ora_robot *, const double *, const double *, int, int, ora_lin *, double *, int);
int ora_linearize_cast(void *, const ora_robot *, const double *, const double *, int, int, ora_lin *, double *, int);
void *emu_create(const ora_scene *);
void emu_destroy(void *);
void emu_set_contact_distance(void *, float);
int emu_links_vs_all(void *, const double *, const int *, int, int, ora_contact *, int);
int emu_cast_vs_all(void *, const double *, const int *, int, const double *, const double *, ora_contact *, int);
int emu_continuous_check(void *, const double *, const int *, int, const double *, int, ora_sweep_hit *, int);
int emu_ray_cast(void *, const double *, const double *, const double *, int, float *);
}

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double urand() {   // xorshift64*: [0, 1)
  g_s ^= g_s >> 12; g_s ^= g_s << 25; g_s ^= g_s >> 27;
  return (double)((g_s * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double uni(double a, double b) { return a + (b - a) * urand(); }
static void randT7(double *t, double spread) {
  double q[4], n = 0;
  for (int k = 0; k < 4; ++k) q[k] = uni(-1, 1), n += q[k] * q[k];
  n = std::sqrt(n);
  for (int k = 0; k < 4; ++k) t[k] = q[k] / n;
  for (int k = 0; k < 3; ++k) t[4 + k] = uni(-spread, spread);
}

struct Scene {
  std::vector<int> kind, voff, vcnt, child_shape, off, cnt, compound, group;
  std::vector<float> param, verts;
  std::vector<double> child_T, objT;
  std::vector<int> robot;
  ora_scene view() const {
    ora_scene s;
    s.n_shapes = (int)kind.size(); s.shape_kind = kind.data(); s.shape_param = param.data(); s.shape_voff = voff.data(); s.shape_vcnt = vcnt.data();
    s.n_verts = (int)verts.size() / 3; s.verts = verts.data();
    s.n_children = (int)child_shape.size(); s.child_shape = child_shape.data(); s.child_T = child_T.data();
    s.n_objects = (int)off.size(); s.obj_child_off = off.data(); s.obj_child_cnt = cnt.data(); s.obj_compound = compound.data(); s.obj_group = group.data();
    return s;
  }
};
static void addShape(Scene &s) {
  const double r = urand();
  float p[4] = {0, 0, 0, 0};
  s.voff.push_back((int)s.verts.size() / 3);
  if (r < 0.3) { s.kind.push_back(ORA_BOX); for (int k = 0; k < 3; ++k) p[k] = (float)uni(0.03, 0.25); }
  else if (r < 0.45) { s.kind.push_back(ORA_SPHERE); p[0] = (float)uni(0.03, 0.2); }
  else if (r < 0.6) { s.kind.push_back(ORA_CYLZ); p[0] = (float)uni(0.03, 0.15); p[1] = (float)uni(0.03, 0.25); }
  else {
    s.kind.push_back(ORA_HULL);
    const int n = 4 + (int)(urand() * 60.99);   // up to 64 points: the vertex-array limits
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) s.verts.push_back((float)uni(-0.15, 0.15));
  }
  s.vcnt.push_back((int)s.verts.size() / 3 - s.voff.back());
  s.param.insert(s.param.end(), p, p + 4);
}
static void addObject(Scene &s, int group, double spread) {
  const int k = urand() < 0.5 ? 1 : 1 + (int)(urand() * 2.99);
  const bool bare = k == 1 && urand() < 0.6;
  s.off.push_back((int)s.child_shape.size()); s.cnt.push_back(k); s.compound.push_back(bare ? 0 : 1); s.group.push_back(group);
  for (int i = 0; i < k; ++i) {
    addShape(s);
    s.child_shape.push_back((int)s.kind.size() - 1);
    double t[7] = {1, 0, 0, 0, 0, 0, 0};
    if (!bare) randT7(t, 0.1);
    s.child_T.insert(s.child_T.end(), t, t + 7);
  }
  double T[7];
  randT7(T, spread);
  s.objT.insert(s.objT.end(), T, T + 7);
}

static long g_fail = 0;
#define CHECK(c) do { if (!(c)) { ++g_fail; fprintf(stderr, "san_driver: CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); } } while (0)

static bool sameContacts(std::vector<ora_contact> a, std::vector<ora_contact> b) {
  if (a.size() != b.size()) return false;
  for (const ora_contact &x : a) {   // as sets: the two walk the pairs in the same order, but do not rely on it
    bool found = false;
    for (const ora_contact &y : b)
      if (x.objA == y.objA && x.childA == y.childA && x.objB == y.objB && x.childB == y.childB && x.cctype == y.cctype && x.time == y.time &&
          x.dist == y.dist && !memcmp(x.n, y.n, sizeof x.n) && !memcmp(x.ptA, y.ptA, sizeof x.ptA) && !memcmp(x.ptB, y.ptB, sizeof x.ptB) &&
          !memcmp(x.ptB1, y.ptB1, sizeof x.ptB1)) { found = true; break; }
    if (!found) return false;
  }
  return true;
}

int main(int argc, char **argv) {
  const int nscenes = argc > 1 ? atoi(argv[1]) : 24;
  long contacts = 0, hits = 0, rayhits = 0, rows = 0;
  for (int sc = 0; sc < nscenes; ++sc) {
    Scene s;
    const double spread = sc % 3 == 0 ? 0.3 : (sc % 3 == 1 ? 0.6 : 1.2);
    for (int i = 0; i < 7; ++i) addObject(s, ORA_GROUP_KINBODY, spread);
    for (int i = 0; i < 5; ++i) { addObject(s, ORA_GROUP_ROBOT, spread); s.robot.push_back((int)s.off.size() - 1); }
    const ora_scene v = s.view();
    void *wo = ora_create(&v), *we = emu_create(&v);
    const int nq = (int)s.robot.size(), cap = 4096;
    std::vector<double> T0(7 * nq), T1(7 * nq);
    for (int i = 0; i < nq; ++i) {
      memcpy(&T0[7 * i], &s.objT[7 * s.robot[i]], 7 * sizeof(double));
      randT7(&T1[7 * i], spread);
      if (i % 2) for (int k = 0; k < 4; ++k) T1[7 * i + k] = T0[7 * i + k];   // every other link only translates
    }
    for (float d : {0.0f, 0.09f}) {
      ora_set_contact_distance(wo, d); emu_set_contact_distance(we, d);
      std::vector<ora_contact> a(cap), b(cap);
      int na = ora_links_vs_all(wo, s.objT.data(), s.robot.data(), nq, -1, a.data(), cap), nb = emu_links_vs_all(we, s.objT.data(), s.robot.data(), nq, -1, b.data(), cap);
      CHECK(na == nb && na <= cap);
      a.resize(na < cap ? na : cap); b.resize(nb < cap ? nb : cap);
      CHECK(sameContacts(a, b));
      contacts += na;
      a.assign(cap, ora_contact()); b.assign(cap, ora_contact());
      na = ora_cast_vs_all(wo, s.objT.data(), s.robot.data(), nq, T0.data(), T1.data(), a.data(), cap);
      nb = emu_cast_vs_all(we, s.objT.data(), s.robot.data(), nq, T0.data(), T1.data(), b.data(), cap);
      CHECK(na == nb && na <= cap);
      a.resize(na < cap ? na : cap); b.resize(nb < cap ? nb : cap);
      CHECK(sameContacts(a, b));
      contacts += na;
      // ContinuousCheckTrajectory over 3 segments
      const int nst = 4;
      std::vector<double> traj((size_t)nst * nq * 7);
      for (int t = 0; t < nst; ++t) for (int i = 0; i < nq; ++i) randT7(&traj[((size_t)t * nq + i) * 7], spread);
      std::vector<ora_sweep_hit> ha(cap), hb(cap);
      const int ma = ora_continuous_check(wo, s.objT.data(), s.robot.data(), nq, traj.data(), nst, ha.data(), cap);
      const int mb = emu_continuous_check(we, s.objT.data(), s.robot.data(), nq, traj.data(), nst, hb.data(), cap);
      CHECK(ma == mb && ma <= cap);
      for (int i = 0; i < ma && i < mb; ++i)
        CHECK(ha[i].obj == hb[i].obj && ha[i].child == hb[i].child && ha[i].seg == hb[i].seg && ha[i].hitObj == hb[i].hitObj && ha[i].hitChild == hb[i].hitChild &&
              ha[i].fraction == hb[i].fraction && !memcmp(ha[i].n, hb[i].n, sizeof ha[i].n) && !memcmp(ha[i].pt, hb[i].pt, sizeof ha[i].pt));
      hits += ma;
      // RayCastCollision
      const int nr = 200;
      std::vector<double> from(3 * nr), to(3 * nr);
      for (int i = 0; i < 3 * nr; ++i) from[i] = uni(-1.5 * spread, 1.5 * spread), to[i] = uni(-1.5 * spread, 1.5 * spread);
      for (int k = 0; k < 3; ++k) to[k] = from[k];   // a zero-length ray
      std::vector<float> fa(nr), fb(nr);
      ora_ray_cast(wo, s.objT.data(), from.data(), to.data(), nr, fa.data());
      emu_ray_cast(we, s.objT.data(), from.data(), to.data(), nr, fb.data());
      CHECK(!memcmp(fa.data(), fb.data(), nr * sizeof(float)));
      for (float f : fa) rayhits += f < 1.0f;
    }
    // empty query, the oracle's capacity handling (cap smaller than the result)
    {
      ora_contact one[1];
      CHECK(ora_links_vs_all(wo, s.objT.data(), s.robot.data(), 0, -1, one, 1) == 0);
      ora_set_contact_distance(wo, 0.3f);
      const int n = ora_links_vs_all(wo, s.objT.data(), s.robot.data(), nq, -1, one, 1);
      CHECK(n >= 0);
    }
    // FK + linearization (oracle only; the device's runs on the GPU): a 3-link arm carrying three of the robot objects
    {
      const int nl = 4;
      int parent[nl] = {-1, 0, 1, 2}, jtype[nl] = {ORA_J_FIXED, ORA_J_REVOLUTE, ORA_J_PRISMATIC, ORA_J_REVOLUTE}, dof[nl] = {-1, 0, 1, 2};
      int link_obj[nl] = {-1, s.robot[0], s.robot[1], s.robot[2]}, joint_inds[3] = {0, 1, 2};
      double origin[nl * 7], axis[nl * 3] = {0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 1, 0}, mult[nl] = {1, 1, 0.5, -1}, offs[nl] = {0, 0.1, 0, 0.2}, q[3] = {0, 0, 0};
      for (int i = 0; i < nl; ++i) { double t[7] = {1, 0, 0, 0, 0.2 * i, 0.05 * i, 0}; memcpy(origin + 7 * i, t, sizeof t); }
      ora_robot rb;
      memset(&rb, 0, sizeof rb);
      rb.n_links = nl; rb.parent = parent; rb.jtype = jtype; rb.T_origin = origin; rb.axis = axis; rb.dof = dof; rb.mult = mult; rb.offs = offs;
      rb.link_obj = link_obj; rb.n_dof = 3; rb.q_robot = q; rb.base_T[0] = 1; rb.n_joint_inds = 3; rb.joint_inds = joint_inds;
      rb.affine = ORA_DOF_X | ORA_DOF_ROTAXIS; rb.rot_axis[2] = 1;
      const int K = ora_robot_dof(&rb), T = 5;
      CHECK(K == 5);
      std::vector<double> dofs((size_t)T * K);
      for (double &x : dofs) x = uni(-0.8, 0.8);
      std::vector<ora_lin> lin(cap);
      std::vector<double> grads((size_t)cap * 2 * K);
      ora_set_contact_distance(wo, 0.15f);
      const int n1 = ora_linearize_discrete(wo, &rb, s.objT.data(), dofs.data(), T, -1, lin.data(), grads.data(), cap);
      for (int i = 0; i < n1 && i < cap; ++i) {   // the linearization is exact at x: const + grad . x == dist
        double v = lin[i].const0;
        for (int k = 0; k < K; ++k) v += grads[(size_t)i * 2 * K + k] * dofs[(size_t)lin[i].t * K + k];
        CHECK(std::fabs(v - lin[i].c.dist) < 1e-9);
      }
      const int n2 = ora_linearize_cast(wo, &rb, s.objT.data(), dofs.data(), T, 1, lin.data(), grads.data(), cap);
      CHECK(n1 >= 0 && n2 >= 0);
      rows += n1 + n2;
    }
    ora_destroy(wo); emu_destroy(we);
  }
  printf("san_driver: %d scenes, %ld contacts, %ld sweep hits, %ld ray hits, %ld linearized rows; %ld failed checks\n", nscenes, contacts, hits, rayhits, rows, g_fail);
  if (g_fail || contacts < 100 || hits < 10 || rayhits < 100 || rows < 10) { printf("san_driver: FAILED\n"); return 1; }
  printf("san_driver: OK\n");
  return 0;
}
