// host_unit.cpp -- tests of the C++ host layer (trajopt_amd/host) through the C-ABI, on a real MI355X.
//
//   host_unit box_distance        the reference's TEST(collision_checker, box_distance), collision-checker-unit.cpp:29-90
//   host_unit continuous_boxes    the reference's TEST(continuous_collisions, boxes),    collision-checker-unit.cpp:102-153
//   host_unit linearize IN OUT    one SQP evaluation of a problem with a discrete and a continuous collision term per
//                                 timestep (problem_description.cpp:516-529) on a world + trajectory written by
//                                 tests/util.py:write_flat_world: the driver below asks the checker what the reference's
//                                 collision terms ask it (LinksVsAll per timestep, CastVsAll per pair), once call by call
//                                 and once behind a CollisionPrefetchCost; every contact and hinge row is dumped to OUT,
//                                 where pytest compares them with the Python host and the oracle.
//   host_unit fk_guard            UploadRobot refuses a kinematic tree whose device FK does not reproduce GetTransform()
//   host_unit grab_rebind         Grab / Release between two evaluators on the same robot and joints
// The environment is built from plain structs (rave_shim.hpp) because OpenRAVE and its XML loader are absent; the box
// bodies below are data/box.xml and data/boxbot.xml (a 1 m cube each).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "collision_prefetch_cost.hpp"
#include "hip_collision_checker.hpp"
#include "rave_access.hpp"   // (the stand-in's: TestTamper)

using namespace OR;
using namespace trajopt;
using std::string;

static int g_failures = 0;
#define EXPECT_TRUE(c) \
  do { if (!(c)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_failures; } } while (0)
#define EXPECT_EQ(a, b) \
  do { if (!((a) == (b))) { std::cout << "FAIL " << __FILE__ << ":" << __LINE__ << ": " #a " == " #b " (" << (a) << " vs " << (b) << ")\n"; ++g_failures; } } while (0)
#define EXPECT_NEAR(a, b, tol) \
  do { if (!(fabs((double)(a) - (double)(b)) <= (tol))) { std::cout << "FAIL " << __FILE__ << ":" << __LINE__ << ": |" #a " - " #b "| <= " #tol " (" << (a) << " vs " << (b) << ")\n"; ++g_failures; } } while (0)
// the reference's macro compares x three times (collision-checker-unit.cpp:92-100); this one compares x, y and z
#define EXPECT_VECTOR_NEAR(v0, v1, tol) \
  do { Vector a_ = (v0), b_ = (v1); EXPECT_NEAR(a_.x, b_.x, tol); EXPECT_NEAR(a_.y, b_.y, tol); EXPECT_NEAR(a_.z, b_.z, tol); } while (0)

static KinBodyPtr makeBox(const string& name) {   // data/box.xml
  KinBodyPtr b(new KinBody(name));
  KinBody::LinkPtr l = b->AddLink(name);
  KinBody::Link::Geometry g;
  g.type = GT_Box, g.boxExtents = Vector(.5, .5, .5);
  l->AddGeometry(g);
  return b;
}
static RobotBasePtr makeBoxbot() {   // data/boxbot.xml
  RobotBasePtr r(new RobotBase("boxbot"));
  KinBody::LinkPtr l = r->AddLink("boxlink", RobotBase::Joint());
  KinBody::Link::Geometry g;
  g.type = GT_Box, g.boxExtents = Vector(.5, .5, .5);
  l->AddGeometry(g);
  return r;
}

static void test_box_distance() {
  EnvironmentBasePtr env(new EnvironmentBase);
  KinBodyPtr box0 = makeBox("box0"), box1 = makeBox("box1");
  env->Add(box0), env->Add(box1);
  CollisionCheckerPtr checker = CreateCollisionChecker(env);

#define EXPECT_NUM_COLLISIONS(n)                             \
  {                                                          \
    vector<Collision> allvsall_collisions, bodyvsall_collisions; \
    checker->AllVsAll(allvsall_collisions);                  \
    EXPECT_EQ((size_t)(n), allvsall_collisions.size());      \
    checker->BodyVsAll(*box0, bodyvsall_collisions);         \
    EXPECT_EQ((size_t)(n), bodyvsall_collisions.size());     \
  }
  // (the reference comments the AllVsAll count out, :44: its manifolds remember the contact distance they were created
  // under, so the count depends on the history of SetContactDistance calls (:464 "seems to be broken").  The engine's
  // AllVsAll is that of a checker whose distance was set before the first call: contacts up to 2.001 d, which leaves
  // every count of this test as BodyVsAll's)
  checker->SetContactDistance(0);
  box1->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(.9, 0, 0)));
  box0->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(0, 0, 0)));
  EXPECT_NUM_COLLISIONS(1);
  {   // SURVEY.md 8c known answer for this exactly face-parallel overlap: Bullet's GJK -> EPA gives a diagonal normal
    vector<Collision> c;
    checker->BodyVsAll(*box0, c);
    if (c.size() == 1) EXPECT_EQ((float)c[0].distance, -0.0577350222f);
  }
  checker->SetContactDistance(0);
  box1->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(1.1, 0, 0)));
  EXPECT_NUM_COLLISIONS(0);
  checker->SetContactDistance(.04);
  EXPECT_NUM_COLLISIONS(0);
  checker->SetContactDistance(.1);
  box1->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(1.09, 0, 0)));
  EXPECT_NUM_COLLISIONS(1);
  checker->SetContactDistance(.2);
  box1->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(1.1, 0, 0)));
  EXPECT_NUM_COLLISIONS(1);
  {   // Exclude / IncludeCollisionPair (:342-351): one direction, like the reference
    checker->ExcludeCollisionPair(*box0->GetLinks()[0], *box1->GetLinks()[0]);
    vector<Collision> a, b;
    checker->BodyVsAll(*box0, a), checker->BodyVsAll(*box1, b);
    EXPECT_EQ(a.size(), (size_t)0);
    EXPECT_EQ(b.size(), (size_t)1);
    checker->IncludeCollisionPair(*box0->GetLinks()[0], *box1->GetLinks()[0]);
    a.clear();
    checker->BodyVsAll(*box0, a);
    EXPECT_EQ(a.size(), (size_t)1);
  }
  env->Remove(box1);
  EXPECT_NUM_COLLISIONS(0);
  // GetOrCreate caches one checker per environment (collision_checker.cpp:16-27)
  EXPECT_TRUE(CollisionChecker::GetOrCreate(*env) == CollisionChecker::GetOrCreate(*env));
}

static void test_continuous_boxes() {
  EnvironmentBasePtr env(new EnvironmentBase);
  KinBodyPtr box = makeBox("box");
  RobotBasePtr boxbot = makeBoxbot();
  env->Add(box), env->Add(boxbot);
  CollisionCheckerPtr checker = CreateCollisionChecker(env);
  {
    RobotAndDOFPtr rad(new RobotAndDOF(boxbot, IntVec(), DOF_X | DOF_Y, Vector(), env));
    TrajArray traj(2, 2);
    traj(0, 0) = -1.9, traj(0, 1) = 0, traj(1, 0) = 0, traj(1, 1) = 1.9;
    vector<Collision> collisions;
    checker->ContinuousCheckTrajectory(traj, *rad, collisions);
    EXPECT_EQ(collisions.size(), (size_t)1);
    if (collisions.size() == 1) {
      Collision col = collisions[0];
      Vector robot_normal = (col.linkA == boxbot->GetLinks()[0].get() ? 1. : -1.) * col.normalB2A;
      EXPECT_VECTOR_NEAR(robot_normal, Vector(-1, 0, 0), 1e-4);
      EXPECT_EQ((float)col.const1, 0.4736842f);   // time of impact, DESIGN.md 2 (bit-exact vs the reference's convexSweepTest)
    }
  }
  {
    // RayCastCollision (bullet_collision_checker.cpp:352-360): box at the origin (the boxbot, wherever the sweep left it, is not on these rays' first hit);
    // fractions are the reference Bullet's own (tests/test_oracle_cpu.py RAY_CASES)
    EXPECT_TRUE(checker->RayCastCollision(Vector(-2, 0, 0), Vector(2, 0, 0)));
    EXPECT_TRUE(!checker->RayCastCollision(Vector(-2, 0, 3), Vector(2, 0, 3)));     // passes above both boxes
    EXPECT_TRUE(!checker->RayCastCollision(Vector(0, 0, 3), Vector(0, 0, 2.6)));    // ends before them
    EXPECT_TRUE(!checker->RayCastCollision(Vector(0, 0, 0), Vector(0, 0, 3)));      // starts inside: the cast reports nothing
    HIPCollisionChecker* hip = dynamic_cast<HIPCollisionChecker*>(checker.get());
    vector<float> fr = hip->RayCastCollisions({-2, 0, 0, -2, 0.3, 0.1}, {2, 0, 0, 2, -0.2, 0.3});
    EXPECT_EQ(fr[0], 0.375f);
    EXPECT_EQ(fr[1], 0.37474722f);
  }
  struct Case { double q0[2], q1[2]; Vector n; double time, ttol; };
  const double s = 1 / sqrtf(2);
  Case cases[3] = {{{-1.9, 0}, {0, 1.9}, Vector(-s, s, 0), .5, 1e-1}, {{0, .9}, {0, 2}, Vector(0, 1, 0), 0, 1e-6}, {{0, 2}, {0, .9}, Vector(0, 1, 0), 1, 1e-6}};
  for (const Case& c : cases) {
    RobotAndDOFPtr rad(new RobotAndDOF(boxbot, IntVec(), DOF_X | DOF_Y, Vector(), env));
    vector<Collision> collisions;
    checker->CastVsAll(*rad, rad->GetRobot()->GetLinks(), DblVec(c.q0, c.q0 + 2), DblVec(c.q1, c.q1 + 2), collisions);
    EXPECT_EQ(collisions.size(), (size_t)1);
    if (collisions.size() != 1) continue;
    Collision col = collisions[0];
    Vector robot_normal = (col.linkA == boxbot->GetLinks()[0].get() ? 1. : -1.) * col.normalB2A;
    EXPECT_VECTOR_NEAR(robot_normal, c.n, 1e-4);
    EXPECT_NEAR(col.time, c.time, c.ttol);
    EXPECT_TRUE(col.linkB == boxbot->GetLinks()[0].get());   // the swept link is B (bullet_collision_checker.cpp:913-917)
    EXPECT_EQ(col.grad0.size(), (size_t)2);
  }
  {   // CastVsAll on a link list that is not "the affected links": the explicit-pose route, same contact
    RobotAndDOFPtr rad(new RobotAndDOF(boxbot, IntVec(), DOF_X | DOF_Y, Vector(), env));
    vector<KinBody::LinkPtr> links = boxbot->GetLinks();
    links.push_back(box->GetLinks()[0]);
    vector<Collision> a, b;
    double q0[2] = {-1.9, 0}, q1[2] = {0, 1.9};
    checker->CastVsAll(*rad, boxbot->GetLinks(), DblVec(q0, q0 + 2), DblVec(q1, q1 + 2), a);
    checker->CastVsAll(*rad, links, DblVec(q0, q0 + 2), DblVec(q1, q1 + 2), b);
    EXPECT_EQ(a.size(), (size_t)1);
    // the box link as a (resting) cast query: its swept shape is a temporary object outside the world, so contactTest
    // skips nothing and it meets its own object, which passes the KinBodyFilter mask -- the reference does the same
    // (bullet_collision_checker.cpp:974-998, 886-892); robot links never do (RobotFilter is not in the mask)
    EXPECT_EQ(b.size(), (size_t)2);
    if (a.size() == 1 && b.size() == 2) {
      const int k = (b[0].linkA == b[0].linkB) ? 1 : 0;
      EXPECT_TRUE(a[0].distance == b[k].distance && a[0].time == b[k].time && a[0].normalB2A.x == b[k].normalB2A.x && a[0].normalB2A.y == b[k].normalB2A.y);
      EXPECT_TRUE(b[k].grad0.empty());
      EXPECT_TRUE(b[1 - k].linkA == box->GetLinks()[0].get() && b[1 - k].linkB == box->GetLinks()[0].get() && b[1 - k].distance < 0);
    }
  }
}

// ---- flat world files (tests/util.py:write_flat_world) --------------------------------------------------------------
typedef std::map<string, vector<double> > Flat;
static Flat readFlat(const string& path) {
  std::ifstream f(path.c_str());
  if (!f) throw std::runtime_error("cannot open " + path);
  Flat out;
  string name;
  size_t n;
  while (f >> name >> n) {
    vector<double>& v = out[name];
    v.resize(n);
    string tok;
    for (size_t i = 0; i < n; ++i) f >> tok, v[i] = strtod(tok.c_str(), NULL);
  }
  return out;
}
static Transform T7(const vector<double>& v, size_t i) { return Transform(Vector(v[7 * i], v[7 * i + 1], v[7 * i + 2], v[7 * i + 3]), Vector(v[7 * i + 4], v[7 * i + 5], v[7 * i + 6])); }

struct World {
  EnvironmentBasePtr env;
  RobotBasePtr robot;
  RobotAndDOFPtr rad;
  TrajArray traj;
  double d, dist_pen, coeff;
  struct Grab { KinBodyPtr body; int link; Transform rel; };
  vector<Grab> grabs;
};

static void addGeoms(KinBody::Link& link, const Flat& F, int obj) {
  const vector<double>&off = F.at("obj_off"), &cnt = F.at("obj_cnt"), &cs = F.at("child_shape"), &kind = F.at("kind"), &param = F.at("param"),
                      &voff = F.at("voff"), &vcnt = F.at("vcnt"), &verts = F.at("verts"), &cT = F.at("child_T");
  for (int c = (int)off[obj]; c < (int)(off[obj] + cnt[obj]); ++c) {
    int s = (int)cs[c];
    KinBody::Link::Geometry g;
    g.t = T7(cT, c);
    switch ((int)kind[s]) {
      case TOHC_BOX: g.type = GT_Box, g.boxExtents = Vector(param[4 * s], param[4 * s + 1], param[4 * s + 2]); break;
      case TOHC_CYLZ: g.type = GT_Cylinder, g.radius = param[4 * s], g.height = 2 * param[4 * s + 1]; break;
      case TOHC_SPHERE: g.type = GT_Sphere, g.radius = param[4 * s]; break;
      case TOHC_HULL: {
        g.type = GT_TriMesh;
        const vector<double>&mvoff = F.at("mesh_voff"), &mvcnt = F.at("mesh_vcnt"), &mtoff = F.at("mesh_toff"), &mtcnt = F.at("mesh_tcnt"),
                            &mverts = F.at("mesh_verts"), &mtris = F.at("mesh_tris");
        if (mvcnt[s] > 0) {   // the geometry's own mesh (Scene.add_trimesh): the checker reduces it
          for (int v = (int)mvoff[s]; v < (int)(mvoff[s] + mvcnt[s]); ++v) g.mesh.vertices.push_back(Vector(mverts[3 * v], mverts[3 * v + 1], mverts[3 * v + 2]));
          for (int t = 3 * (int)mtoff[s]; t < 3 * (int)(mtoff[s] + mtcnt[s]); ++t) g.mesh.indices.push_back((int)mtris[t]);
        } else {              // hull points given directly: a strip of triangles over them (fewer than 50: used as they are)
          for (int v = (int)voff[s]; v < (int)(voff[s] + vcnt[s]); ++v) g.mesh.vertices.push_back(Vector(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]));
          if ((int)vcnt[s] >= 50) throw std::runtime_error("flat world: hulls of 50 points or more need their mesh");
          for (int v = 0; v + 2 < std::max((int)vcnt[s], 3); ++v)
            for (int k = 0; k < 3; ++k) g.mesh.indices.push_back(std::min(v + k, (int)vcnt[s] - 1));
        }
        break;
      }
      default: throw std::runtime_error("flat world: unsupported shape kind");
    }
    link.AddGeometry(g);
  }
  const Transform& g0 = link.m_geoms[0]->t;
  bool compound = !(link.m_geoms.size() == 1 && fabs(g0.rot.x - 1) < 1e-6 && fabs(g0.rot.y) < 1e-6 && fabs(g0.rot.z) < 1e-6 && fabs(g0.rot.w) < 1e-6 &&
                    fabs(g0.trans.x) < 1e-6 && fabs(g0.trans.y) < 1e-6 && fabs(g0.trans.z) < 1e-6);
  if (compound != (F.at("obj_compound")[obj] != 0)) throw std::runtime_error("flat world: object's compound flag is not what the reference's rule gives");
}

static string oname(int o) {
  std::ostringstream s;
  s << "o" << o;
  return s.str();
}

static World loadWorld(const string& path) {
  Flat F = readFlat(path);
  World W;
  W.env.reset(new EnvironmentBase);
  const vector<double>&group = F.at("obj_group"), &staticT = F.at("static_T"), &linkObj = F.at("link_obj");
  int nObj = (int)group.size(), nLinks = (int)linkObj.size();
  vector<int> objLink(nObj, -1);
  const vector<double>& dofOfLink = F.at("dof");
  for (int i = 0; i < nLinks; ++i)
    if (linkObj[i] >= 0 && (int)dofOfLink[i] != TOHC_DOF_GRABBED) objLink[(int)linkObj[i]] = i;   // a grabbed body's link: a KinBody of its own
  W.robot.reset(new RobotBase("robot"));
  bool robotAdded = false;
  for (int o = 0; o < nObj; ++o) {
    if (objLink[o] >= 0) {
      if (!robotAdded) W.env->Add(W.robot), robotAdded = true;
      continue;
    }
    KinBodyPtr b(new KinBody(oname(o)));
    KinBody::LinkPtr l = b->AddLink(oname(o));
    addGeoms(*l, F, o);
    b->SetTransform(T7(staticT, o));
    W.env->Add(b);
  }
  const vector<double>&parent = F.at("parent"), &jtype = F.at("jtype"), &origin = F.at("T_origin"), &axis = F.at("axis"), &dof = F.at("dof"),
                      &mult = F.at("mult"), &offs = F.at("offs");
  int nRobotLinks = nLinks;
  for (int i = 0; i < nLinks; ++i)
    if ((int)dof[i] == TOHC_DOF_GRABBED) { nRobotLinks = std::min(nRobotLinks, i); }
  for (int i = nRobotLinks; i < nLinks; ++i) {   // model.RobotModel.grab(): robot->Grab(body, link) at the given relative pose
    if ((int)dof[i] != TOHC_DOF_GRABBED) throw std::runtime_error("flat world: grabbed links come after the robot's own");
    W.grabs.push_back(World::Grab{W.env->GetKinBody(oname((int)linkObj[i])), (int)parent[i], T7(origin, i)});
  }
  for (int i = 0; i < nRobotLinks; ++i) {
    RobotBase::Joint j;
    j.parent = (int)parent[i], j.type = (JointType)(int)jtype[i], j.origin = T7(origin, i), j.axis = Vector(axis[3 * i], axis[3 * i + 1], axis[3 * i + 2]);
    j.dof = (int)dof[i], j.mult = mult[i], j.offs = offs[i];
    std::ostringstream nm;
    nm << "l" << i;
    KinBody::LinkPtr l = W.robot->AddLink(linkObj[i] >= 0 ? oname((int)linkObj[i]) : nm.str(), j);
    if (linkObj[i] >= 0) addGeoms(*l, F, (int)linkObj[i]);
  }
  int nDof = (int)F.at("n_dof")[0];
  DblVec q(F.at("q_robot").begin(), F.at("q_robot").begin() + nDof);
  while ((int)W.robot->GetDOF() < nDof) throw std::runtime_error("flat world: n_dof exceeds the joints' dof indices");
  W.robot->SetDOFValues(q);
  W.robot->SetTransform(T7(F.at("base_T"), 0));
  IntVec ji;
  for (double v : F.at("joint_inds")) ji.push_back((int)v);
  ji.resize((size_t)F.at("n_joint_inds")[0]);
  const vector<double>& ra = F.at("rot_axis");
  W.rad.reset(new RobotAndDOF(W.robot, ji, (int)F.at("affine")[0], Vector(ra[0], ra[1], ra[2]), W.env));
  int K = W.rad->GetDOF(), T = (int)(F.at("traj").size() / K);
  W.traj = TrajArray(T, K);
  W.traj.m_data = F.at("traj");
  W.d = F.at("d")[0], W.dist_pen = F.at("dist_pen")[0], W.coeff = F.at("coeff")[0];
  return W;
}

// ---- the driver: one SQP evaluation of the problem's collision terms ---------------------------------------------------
// A collision term of the problem: the discrete cost of timestep t, or the continuous cost of the pair (t, t + 1)
// (CollisionCostInfo::hatch makes one of each kind per timestep, problem_description.cpp:516-529).  The reference's term
// objects live in collision_terms.cpp, which upstream stays as it is; what this driver restates is only WHAT THEY ASK the
// checker and what they make of the answer -- one row per contact,
//     discrete:  dist(x) ~ const0 + grad0 . x_t
//     cast:      dist(x) ~ (1 - time) (const0 + grad0 . x_t) + time (const1 + grad1 . x_{t+1})      (collision_terms.cpp:64-80)
// turned into the hinge argument dist_pen - dist(x) (:205-214) and the value sum coeff max(0, dist_pen - dist) (:215-223).
struct Term { bool cast; int t; };
struct Row {   // constant + sum coeffs[i] * x[vars[i]]
  double constant;
  vector<int> vars;
  DblVec coeffs;
  double at(const DblVec& x) const {
    double v = constant;
    for (size_t i = 0; i < vars.size(); ++i) v += coeffs[i] * x[vars[i]];
    return v;
  }
};

struct TermDriver {
  CollisionCheckerPtr cc;
  RobotAndDOFPtr rad;
  vector<KinBody::LinkPtr> links;   // GetAffectedLinks(only_with_geom): what the evaluators query (collision_terms.cpp:100-115)
  int K;
  double dist_pen, coeff;
  TermDriver(World& W) : cc(CollisionChecker::GetOrCreate(*W.env)), rad(W.rad), K(W.traj.cols()), dist_pen(W.dist_pen), coeff(W.coeff) {
    vector<int> inds;
    rad->GetAffectedLinks(links, true, inds);
  }
  DblVec dofs(const DblVec& x, int t) const { return DblVec(x.begin() + (size_t)t * K, x.begin() + (size_t)(t + 1) * K); }
  // the checker call of one term (collision_terms.cpp:118-122 / :156-161)
  vector<Collision> contacts(const Term& m, const DblVec& x) const {
    vector<Collision> out;
    if (m.cast) {
      rad->SetDOFValues(dofs(x, m.t));
      cc->CastVsAll(*rad, links, dofs(x, m.t), dofs(x, m.t + 1), out);
    } else {
      cc->BindConfiguration(*rad);   // the one line a maintainer adds to SingleTimestepCollisionEvaluator
      rad->SetDOFValues(dofs(x, m.t));
      cc->LinksVsAll(links, out, -1);
    }
    return out;
  }
  double value(const vector<Collision>& cs) const {
    double v = 0;
    for (const Collision& c : cs) v += std::max(0.0, dist_pen - c.distance) * coeff;
    return v;
  }
  // hinge argument dist_pen - dist(x) of every contact, from the linearization the checker put into the Collision
  vector<Row> hinges(const Term& m, const vector<Collision>& cs) const {
    vector<Row> out;
    for (const Collision& c : cs) {
      if ((int)c.grad0.size() != K || (m.cast && (int)c.grad1.size() != K)) throw std::runtime_error("contact without a linearization");
      Row r;
      const float w0f = m.cast ? 1 - c.time : 1.0f;   // `1 - collisions[i].time` is float arithmetic in the reference (:73)
      const double w0 = w0f, w1 = m.cast ? (double)c.time : 0.0;
      r.constant = dist_pen - (m.cast ? w0 * c.const0 + w1 * c.const1 : c.const0);
      for (int k = 0; k < K; ++k) r.vars.push_back(m.t * K + k), r.coeffs.push_back(m.cast ? -(w0 * c.grad0[k]) : -c.grad0[k]);
      if (m.cast)
        for (int k = 0; k < K; ++k) r.vars.push_back((m.t + 1) * K + k), r.coeffs.push_back(-(w1 * c.grad1[k]));
      out.push_back(r);
    }
    return out;
  }
};

struct Pass {
  vector<vector<Collision> > contacts;   // per term
  vector<vector<Row> > hinges;
  DblVec values;
};

// evaluateCosts then convexifyCosts over the cost list, in order (optimizers.cpp:36-56).  With `prefetch` the list starts
// with a CollisionPrefetchCost, which is all it takes: nothing below calls PrefetchTrajectory.
static Pass runCosts(World& W, const DblVec& x, bool prefetch, long* prefetchCalls = NULL) {
  const int T = W.traj.rows(), K = W.traj.cols();
  TermDriver D(W);
  vector<Term> terms;
  for (int t = 0; t < T; ++t) terms.push_back(Term{false, t});
  for (int t = 0; t + 1 < T; ++t) terms.push_back(Term{true, t});
  std::shared_ptr<CollisionPrefetchCost> pf;
  if (prefetch) {
    vector<VarVector> rows(T, VarVector(K));
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < K; ++k) rows[t][k] = Var(t * K + k);
    pf.reset(new CollisionPrefetchCost(W.rad, rows, true, true));
  }
  Pass P;
  Model model;
  if (pf) EXPECT_TRUE(pf->value(x) == 0);                         // evaluateCosts: the prefetch cost comes first ...
  for (const Term& m : terms) P.values.push_back(D.value(D.contacts(m, x)));
  if (pf) EXPECT_TRUE(pf->convex(x, &model)->hinges.empty());     // convexifyCosts: ... and again (same x: no second GPU call)
  for (const Term& m : terms) {
    P.contacts.push_back(D.contacts(m, x));
    P.hinges.push_back(D.hinges(m, P.contacts.back()));
  }
  if (pf && prefetchCalls) *prefetchCalls = pf->NumPrefetches();
  return P;
}

static void dumpRow(FILE* f, const Row& e) {
  fprintf(f, " %.17g %d", e.constant, (int)e.coeffs.size());
  for (size_t i = 0; i < e.coeffs.size(); ++i) fprintf(f, " %d %.17g", e.vars[i], e.coeffs[i]);
}

static void test_linearize(const string& in, const string& out) {
  World W = loadWorld(in);
  CollisionCheckerPtr checker = CollisionChecker::GetOrCreate(*W.env);
  HIPCollisionChecker* hip = dynamic_cast<HIPCollisionChecker*>(checker.get());
  EXPECT_TRUE(hip != NULL);
  checker->SetContactDistance((float)W.d);
  for (const World::Grab& g : W.grabs) W.robot->GrabAt(g.body, W.robot->GetLinks()[g.link], g.rel);   // after the bodies were added, like a caller would
  int T = W.traj.rows(), K = W.traj.cols();
  DblVec x = W.traj.m_data;
  const int nTerms = 2 * T - 1;

  long calls0 = hip->NumEngineCalls();
  Pass A = runCosts(W, x, false);
  long callsA = hip->NumEngineCalls() - calls0, hits0 = hip->NumPrefetchHits(), pfCalls = 0;
  Pass B = runCosts(W, x, true, &pfCalls);
  long callsB = hip->NumEngineCalls() - calls0 - callsA, hitsB = hip->NumPrefetchHits() - hits0;
  // call by call: every term asks the checker in the evaluate pass and again in the convexify pass (the reference's
  // evaluators memoise the second, cache.hxx; this driver does not), + the FK guard of the one robot upload
  EXPECT_EQ(callsA, (long)(2 * nTerms + 1));
  EXPECT_EQ(callsB, 1L);                  // behind the prefetch cost: the whole trajectory in ONE engine call ...
  EXPECT_EQ(pfCalls, 1L);                 // ... made by its value(x); its convex(x) found the slab current
  EXPECT_EQ(hitsB, (long)(2 * nTerms));   // and every term's two queries were answered from that slab
  // prefetched == call by call.  Joint DOFs only: bit for bit.  With affine DOFs the call-by-call LinksVsAll reads the base
  // angle back from the body transform (RaveGetAffineDOFValuesFromTransform, ~1e-16 off what was set), the prefetch has
  // the exact vector: equal to that noise.
  const bool exact = W.rad->affinedofs == 0;
  EXPECT_EQ(A.values.size(), B.values.size());
  for (size_t i = 0; i < A.values.size(); ++i) {
    EXPECT_TRUE(exact ? A.values[i] == B.values[i] : fabs(A.values[i] - B.values[i]) <= 1e-9 * (1 + fabs(B.values[i])));
    EXPECT_EQ(A.hinges[i].size(), B.hinges[i].size());
    for (size_t h = 0; h < A.hinges[i].size() && h < B.hinges[i].size(); ++h) {
      const Row &a = A.hinges[i][h], &b = B.hinges[i][h];
      if (exact || (int)i >= T) EXPECT_TRUE(a.constant == b.constant && a.coeffs == b.coeffs);   // (CastVsAll is handed its vectors: exact either way)
      else {
        EXPECT_NEAR(a.constant, b.constant, 1e-9);
        for (size_t k = 0; k < a.coeffs.size() && k < b.coeffs.size(); ++k) EXPECT_NEAR(a.coeffs[k], b.coeffs[k], 1e-9);
      }
    }
  }
  // the same hinges as the device assembles them (tohc_fetch_hinges on the slab the prefetch left) against
  // what sco would hand BPMPD for these rows: ConvexObjective::addHinge (modeling.cpp:22-29: row `viol - h <= 0`) ->
  // BPMPDModel::optimize (bpmpd_interface.cpp:352-369: rhs = -constant; per variable the (row, value) list through
  // simplify2, solver_interface.cpp:23-37: exact zeros dropped, rows ascending).  Bit for bit.
  {
    tohc_hinges H;
    EXPECT_EQ(tohc_fetch_hinges(hip->GetEngine(), W.dist_pen, W.dist_pen, &H), (int)TOHC_OK);
    EXPECT_EQ((int)H.n_problems, 1);
    EXPECT_EQ((int)H.T, T);
    EXPECT_EQ((int)H.K, K);
    // rows of the device block are in canonical order: (t, discrete before cast); the terms come discrete t = 0.., then cast
    size_t nrows = 0;
    for (auto& c : B.hinges) nrows += c.size();
    EXPECT_EQ((size_t)H.n_rows, nrows);
    vector<int> rowOfHinge;   // device row of the h-th hinge in term order
    {
      vector<vector<int> > rowsOf(2 * T);   // [t] discrete, [T + t] cast
      for (int i = 0; i < (int)H.n_rows; ++i) rowsOf[(H.row_t[i] & TOHC_HINGE_CAST) ? T + (H.row_t[i] & ~TOHC_HINGE_CAST) : H.row_t[i]].push_back(i);
      for (size_t ci = 0; ci < B.hinges.size(); ++ci) {
        EXPECT_EQ(rowsOf[ci].size(), B.hinges[ci].size());
        for (int r : rowsOf[ci]) rowOfHinge.push_back(r);
      }
    }
    if ((size_t)H.n_rows == nrows && rowOfHinge.size() == nrows) {
      std::map<int, std::map<int, double> > var2cnt;   // var -> (row -> value), what simplify2 leaves
      size_t h = 0;
      for (auto& c : B.hinges)
        for (const Row& aff : c) {
          const int row = rowOfHinge[h++];
          EXPECT_TRUE(H.rhs[row] == -aff.constant);
          for (size_t j = 0; j < aff.coeffs.size(); ++j)
            if (aff.coeffs[j] != 0) var2cnt[aff.vars[j]][row] += aff.coeffs[j];
        }
      size_t nnz = 0;
      for (int v = 0; v < T * K; ++v) {
        const std::map<int, double>& col = var2cnt[v];
        EXPECT_EQ((size_t)(H.col_off[v + 1] - H.col_off[v]), col.size());
        size_t e = (size_t)H.col_off[v];
        for (auto& rv : col) {
          if (e < (size_t)H.col_off[v + 1]) EXPECT_TRUE(H.col_idx[e] == rv.first && H.col_val[e] == rv.second);
          ++e, ++nnz;
        }
      }
      EXPECT_EQ((size_t)H.nnz, nnz);
      EXPECT_TRUE(nnz > 0);
    }
  }
  // IgnoreZeroStateSelfCollisions() once more, now with a configuration bound (collision_checker.cpp:58-66): the robot is
  // lifted to z = 10 by SetTransform, which the bound DOF vector cannot express -- nothing may change
  // (not with a grabbed body: it is carried along to z = 10, so the call -- rightly -- finds and excludes its contacts
  // with the gripper that holds it)
  if (W.grabs.empty()) {
    checker->IgnoreZeroStateSelfCollisions();
    Pass C = runCosts(W, x, false);
    EXPECT_EQ(A.values.size(), C.values.size());
    for (size_t i = 0; i < A.values.size() && i < C.values.size(); ++i) EXPECT_TRUE(A.values[i] == C.values[i]);
  }
  // the linearization is exact at x: every hinge row evaluates to dist_pen - distance there (cast rows: to the 6e-8 by which
  // the reference's float `1 - time` and `time` miss summing to one)
  for (size_t i = 0; i < B.hinges.size(); ++i)
    for (size_t h = 0; h < B.hinges[i].size(); ++h)
      EXPECT_NEAR(B.hinges[i][h].at(x), W.dist_pen - B.contacts[i][h].distance, 1e-9 + ((int)i >= T ? 1.2e-7 * fabs(B.contacts[i][h].distance) : 0));

  FILE* f = fopen(out.c_str(), "w");
  if (!f) throw std::runtime_error("cannot write " + out);
  fprintf(f, "meta %d %d %ld %ld %ld\n", T, K, callsA, callsB, hitsB);
  for (size_t i = 0; i < B.contacts.size(); ++i) {
    bool cast = (int)i >= T;
    int t = cast ? (int)i - T : (int)i;
    fprintf(f, "cost %s %d %.17g %d\n", cast ? "cast" : "disc", t, B.values[i], (int)B.contacts[i].size());
    for (size_t h = 0; h < B.contacts[i].size(); ++h) {
      const Collision& c = B.contacts[i][h];
      fprintf(f, "contact %s %d %s %d %d %.9g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", c.linkA->GetName().c_str(), c.childA,
              c.linkB->GetName().c_str(), c.childB, (int)c.cctype, c.time, c.distance, c.normalB2A.x, c.normalB2A.y, c.normalB2A.z, c.ptA.x, c.ptA.y, c.ptA.z,
              c.ptB.x, c.ptB.y, c.ptB.z, c.ptB1.x, c.ptB1.y, c.ptB1.z);
      dumpRow(f, B.hinges[i][h]);   // the hinge argument dist_pen - dist(x)
      fprintf(f, "\n");
    }
  }
  // the acceptance check of the reference's planning tests (planning-unit.cpp:120, cast-cost-unit.cpp:75)
  vector<Collision> sweep;
  checker->ContinuousCheckTrajectory(W.traj, *W.rad, sweep);
  for (const Collision& c : sweep)
    fprintf(f, "sweep %s %d %s %d %.9g %.17g %.17g %.17g\n", c.linkA->GetName().c_str(), c.childA, c.linkB->GetName().c_str(), c.childB, c.const1,
            c.normalB2A.x, c.normalB2A.y, c.normalB2A.z);
  fclose(f);
}

// ---- a small arm for the two host-layer tests below: three revolute links with a box each, over a table ----------------
static RobotBasePtr makeArm3() {
  RobotBasePtr r(new RobotBase("arm3"));
  double off = 0;
  for (int i = 0; i < 3; ++i) {
    RobotBase::Joint j;
    j.parent = i - 1, j.type = JointRevolute, j.origin = Transform(Vector(1, 0, 0, 0), Vector(off, 0, i == 0 ? 0.5 : 0)), j.axis = Vector(0, i == 1 ? 1 : 0, i == 1 ? 0 : 1), j.dof = i;
    std::ostringstream nm;
    nm << "arm" << i;
    KinBody::LinkPtr l = r->AddLink(nm.str(), j);
    KinBody::Link::Geometry g;
    g.type = GT_Box, g.boxExtents = Vector(.15, .03, .03), g.t = Transform(Vector(1, 0, 0, 0), Vector(.15, 0, 0));
    l->AddGeometry(g);
    off = 0.3;
  }
  r->SetDOFValues(DblVec{0.2, 0.3, -0.4});
  return r;
}
static KinBodyPtr makeSlab(const string& name, const Vector& half, const Vector& at) {
  KinBodyPtr b(new KinBody(name));
  KinBody::LinkPtr l = b->AddLink(name);
  KinBody::Link::Geometry g;
  g.type = GT_Box, g.boxExtents = half;
  l->AddGeometry(g);
  b->SetTransform(Transform(Vector(1, 0, 0, 0), at));
  return b;
}

// The FK guard (HIPCollisionChecker::CheckDeviceFK): the tree handed to tohc_upload_robot, run through the engine's FK
// kernel, must reproduce every link->GetTransform().  Here the checker is handed a tree that is not the robot's (the
// stand-in's TestTamper hook: what a wrong convention in an implementation of rave_access.hpp would look like) and the
// bind must throw, naming the link; with the true tree it must not.
static void test_fk_guard() {
  EnvironmentBasePtr env(new EnvironmentBase);
  RobotBasePtr arm = makeArm3();
  env->Add(arm), env->Add(makeSlab("table", Vector(.8, .8, .01), Vector(.4, 0, .42)));
  CollisionCheckerPtr checker = CollisionChecker::GetOrCreate(*env);
  checker->SetContactDistance(.1f);
  RobotAndDOFPtr rad(new RobotAndDOF(arm, IntVec{0, 1, 2}, 0, Vector(0, 0, 1), env));
  vector<KinBody::LinkPtr> links;
  vector<int> inds;
  rad->GetAffectedLinks(links, true, inds);
  DblVec q0{0.2, 0.3, -0.4}, q1{0.25, 0.35, -0.3};
  vector<Collision> ok;
  checker->CastVsAll(*rad, links, q0, q1, ok);   // binds, uploads, passes the guard
  EXPECT_TRUE(!ok.empty() && (int)ok[0].grad0.size() == 3);
  auto bindThrows = [&](const IntVec& active, const DblVec& a, const DblVec& b, const char* names) {
    RobotAndDOFPtr r2(new RobotAndDOF(arm, active, 0, Vector(0, 0, 1), env));   // another active set: a new upload
    vector<KinBody::LinkPtr> l2;
    r2->GetAffectedLinks(l2, true, inds);
    try {
      vector<Collision> c;
      checker->CastVsAll(*r2, l2, a, b, c);
    } catch (const std::runtime_error& e) {
      const bool ours = string(e.what()).find("forward kinematics disagrees") != string::npos && (!names || string(e.what()).find(names) != string::npos);
      if (!ours) printf("unexpected exception: %s\n", e.what());
      return ours;
    }
    return false;
  };
  // (i) joint 2 sits 5 mm from where the host's FK has it
  rave_access::TestTamper() = [](vector<rave_access::TreeNode>& n) { n[2].origin.trans.x += 0.005; };
  EXPECT_TRUE(bindThrows(IntVec{0, 1}, DblVec{0.2, 0.3}, DblVec{0.25, 0.35}, "arm2"));
  // (ii) the true tree again: the same bind goes through
  rave_access::TestTamper() = nullptr;
  EXPECT_TRUE(!bindThrows(IntVec{0, 1}, DblVec{0.2, 0.3}, DblVec{0.25, 0.35}, NULL));
  // (iii) a mirrored axis convention (the classic): joint 1 about -y instead of +y
  rave_access::TestTamper() = [](vector<rave_access::TreeNode>& n) { n[1].axis = Vector(0, -1, 0); };
  EXPECT_TRUE(bindThrows(IntVec{1, 2}, DblVec{0.3, -0.4}, DblVec{0.35, -0.3}, "arm1"));
  // (iv) 1e-7 rad of joint offset: above the 1e-9 tolerance at a 0.3 m lever
  rave_access::TestTamper() = [](vector<rave_access::TreeNode>& n) { n[0].offs += 1e-7; };
  EXPECT_TRUE(bindThrows(IntVec{0, 2}, DblVec{0.2, -0.4}, DblVec{0.25, -0.3}, NULL));
  rave_access::TestTamper() = nullptr;
  EXPECT_TRUE(!bindThrows(IntVec{0, 2}, DblVec{0.2, -0.4}, DblVec{0.25, -0.3}, NULL));
}

// bind + LinksVsAll, then robot->Grab, then a NEW evaluator on the same robot and joints: the affected-link list
// now holds the grabbed body's link (configuration_space.cpp:120-131) and the checker must follow -- linearized contacts,
// not the explicit-pose route's bare ones; and back again after Release.
static void test_grab_rebind() {
  EnvironmentBasePtr env(new EnvironmentBase);
  RobotBasePtr arm = makeArm3();
  KinBodyPtr held = makeSlab("held", Vector(.05, .02, .02), Vector(0, 0, 5));
  env->Add(arm), env->Add(makeSlab("table", Vector(.8, .8, .01), Vector(.4, 0, .42))), env->Add(held);
  CollisionCheckerPtr checker = CollisionChecker::GetOrCreate(*env);
  HIPCollisionChecker* hip = dynamic_cast<HIPCollisionChecker*>(checker.get());
  checker->SetContactDistance(.1f);
  DblVec q{0.2, 0.3, -0.4};
  auto query = [&](size_t expectLinks, bool expectHeld) {
    // what a fresh SingleTimestepCollisionEvaluator does (collision_terms.cpp:100-122)
    RobotAndDOFPtr rad(new RobotAndDOF(arm, IntVec{0, 1, 2}, 0, Vector(0, 0, 1), env));
    vector<KinBody::LinkPtr> links;
    vector<int> inds;
    rad->GetAffectedLinks(links, true, inds);
    EXPECT_EQ(links.size(), expectLinks);
    checker->BindConfiguration(*rad);
    rad->SetDOFValues(q);
    vector<Collision> c;
    checker->LinksVsAll(links, c, -1);
    EXPECT_TRUE(!c.empty());
    bool sawHeld = false;
    for (const Collision& col : c) {
      EXPECT_EQ(col.grad0.size(), (size_t)3);   // the linearizing route: every contact carries its row
      sawHeld |= col.linkA == held->GetLinks()[0].get() || col.linkB == held->GetLinks()[0].get();
    }
    EXPECT_TRUE(sawHeld == expectHeld);
    // and the prefetch route follows too
    TrajArray traj(2, 3);
    for (int k = 0; k < 3; ++k) traj(0, k) = q[k], traj(1, k) = q[k] + 0.05;
    hip->PrefetchTrajectory(*rad, traj, true, true);
    long h0 = hip->NumPrefetchHits();
    vector<Collision> c2;
    rad->SetDOFValues(q);
    checker->LinksVsAll(links, c2, -1);
    EXPECT_EQ(hip->NumPrefetchHits(), h0 + 1);
    EXPECT_EQ(c2.size(), c.size());
  };
  query(3, false);
  // the held box: 12 cm past the last link's tip, a few centimetres above the table
  arm->SetDOFValues(q);
  arm->GrabAt(held, arm->GetLinks()[2], Transform(Vector(1, 0, 0, 0), Vector(.36, 0, 0)));
  query(4, true);
  arm->Release(held);
  held->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(0, 0, 5)));
  query(3, false);
}

// CreateCollisionChecker's switch (TRAJOPT_COLLISION_CHECKER = hip | bullet): "bullet" and unknown names are errors, never a
// silent substitute; unset / hip give the HIP checker.  The "bullet" and "unknown" cases need no GPU.
static void test_factory_switch() {
  EnvironmentBasePtr env(new EnvironmentBase);
  const char* bad[2] = {"bullet", "fcl"};
  for (int i = 0; i < 2; ++i) {
    setenv("TRAJOPT_COLLISION_CHECKER", bad[i], 1);
    bool threw = false;
    try { CreateCollisionChecker(env); } catch (const std::runtime_error& e) { threw = string(e.what()).find(bad[i]) != string::npos; }
    EXPECT_TRUE(threw);
  }
  unsetenv("TRAJOPT_COLLISION_CHECKER");
}
static void test_factory_hip() {
  EnvironmentBasePtr env(new EnvironmentBase);
  setenv("TRAJOPT_COLLISION_CHECKER", "hip", 1);
  CollisionCheckerPtr a = CreateCollisionChecker(env);
  unsetenv("TRAJOPT_COLLISION_CHECKER");
  CollisionCheckerPtr b = CreateCollisionChecker(env);
  EXPECT_TRUE(dynamic_cast<HIPCollisionChecker*>(a.get()) != NULL && dynamic_cast<HIPCollisionChecker*>(b.get()) != NULL);
}

int main(int argc, char** argv) {
  string cmd = argc > 1 ? argv[1] : "";
  try {
    if (cmd == "box_distance") test_box_distance();
    else if (cmd == "factory_switch") test_factory_switch();
    else if (cmd == "factory_hip") test_factory_hip();
    else if (cmd == "continuous_boxes") test_continuous_boxes();
    else if (cmd == "linearize" && argc == 4) test_linearize(argv[2], argv[3]);
    else if (cmd == "fk_guard") test_fk_guard();
    else if (cmd == "grab_rebind") test_grab_rebind();
    else { fprintf(stderr, "usage: host_unit box_distance | continuous_boxes | factory_switch | factory_hip | fk_guard | grab_rebind | linearize IN OUT\n"); return 2; }
  } catch (const std::exception& e) {
    printf("EXCEPTION %s\n", e.what());
    return 3;
  }
  printf("%s: %s (%d failures)\n", cmd.c_str(), g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
