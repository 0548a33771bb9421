// host_unit.cpp -- tests of the C++ host layer (trajopt_amd/host) through the C-ABI, on a real MI355X.
//
//   host_unit box_distance        the reference's TEST(collision_checker, box_distance), collision-checker-unit.cpp:29-90
//   host_unit continuous_boxes    the reference's TEST(continuous_collisions, boxes),    collision-checker-unit.cpp:102-153
//   host_unit linearize IN OUT    the collision terms (CollisionCost / CollisionConstraint over SingleTimestep- and
//                                 CastCollisionEvaluator, collision_terms.cpp) on a world + trajectory written by
//                                 tests/util.py:write_flat_world, per-timestep and prefetched; every contact and affine
//                                 expression is dumped to OUT, where pytest compares them with the oracle.
// The environment is built from plain structs (rave_shim.hpp) because OpenRAVE and its XML loader are absent; the box
// bodies below are data/box.xml and data/boxbot.xml (a 1 m cube each).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "collision_terms_standin.hpp"
#include "hip_collision_checker.hpp"

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
  l->m_geoms.push_back(g);
  return b;
}
static RobotBasePtr makeBoxbot() {   // data/boxbot.xml
  RobotBasePtr r(new RobotBase("boxbot"));
  KinBody::LinkPtr l = r->AddLink("boxlink", RobotBase::Joint());
  KinBody::Link::Geometry g;
  g.type = GT_Box, g.boxExtents = Vector(.5, .5, .5);
  l->m_geoms.push_back(g);
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
    link.m_geoms.push_back(g);
  }
  bool compound = !(link.m_geoms.size() == 1 && fabs(link.m_geoms[0].t.rot.x - 1) < 1e-6 && fabs(link.m_geoms[0].t.rot.y) < 1e-6 &&
                    fabs(link.m_geoms[0].t.rot.z) < 1e-6 && fabs(link.m_geoms[0].t.rot.w) < 1e-6 && fabs(link.m_geoms[0].t.trans.x) < 1e-6 &&
                    fabs(link.m_geoms[0].t.trans.y) < 1e-6 && fabs(link.m_geoms[0].t.trans.z) < 1e-6);
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

static void dumpExpr(FILE* f, const AffExpr& e) {
  fprintf(f, " %.17g %d", e.constant, (int)e.coeffs.size());
  for (size_t i = 0; i < e.coeffs.size(); ++i) fprintf(f, " %d %.17g", e.vars[i].var_index, e.coeffs[i]);
}

struct Pass {
  vector<vector<Collision> > contacts;   // per cost
  vector<ConvexObjectivePtr> convex;
  DblVec values;
};

static Pass runCosts(World& W, const DblVec& x, bool prefetch) {
  int T = W.traj.rows(), K = W.traj.cols();
  auto vars = [&](int t) {
    VarVector v(K);
    for (int k = 0; k < K; ++k) v[k] = Var(t * K + k);
    return v;
  };
  vector<std::shared_ptr<CollisionCost> > costs;   // problem_description.cpp:551-566: one cost per timestep / per pair
  for (int t = 0; t < T; ++t) costs.push_back(std::make_shared<CollisionCost>(W.dist_pen, W.coeff, W.rad, vars(t)));
  for (int t = 0; t + 1 < T; ++t) costs.push_back(std::make_shared<CollisionCost>(W.dist_pen, W.coeff, W.rad, vars(t), vars(t + 1)));
  HIPCollisionChecker* hip = dynamic_cast<HIPCollisionChecker*>(CollisionChecker::GetOrCreate(*W.env).get());
  if (prefetch) hip->PrefetchTrajectory(*W.rad, W.traj, true, true);
  Pass P;
  for (auto& c : costs) {
    P.values.push_back(c->value(x));    // evaluateCosts, optimizers.cpp:36-44
    P.convex.push_back(c->convex(x));   // convexifyCosts, :50-56 (hits the evaluator's 3-entry cache)
    vector<Collision> col;
    c->GetEvaluator()->GetCollisionsCached(x, col);
    P.contacts.push_back(col);
  }
  return P;
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

  long calls0 = hip->NumEngineCalls();
  Pass A = runCosts(W, x, false);
  long callsA = hip->NumEngineCalls() - calls0, hits0 = hip->NumPrefetchHits();
  Pass B = runCosts(W, x, true);
  long callsB = hip->NumEngineCalls() - calls0 - callsA, hitsB = hip->NumPrefetchHits() - hits0;
  EXPECT_EQ(callsA, (long)(2 * T - 1));   // one engine call per cost (value -> convex is an evaluator cache hit)
  EXPECT_EQ(callsB, 1L);                  // the whole trajectory in one
  EXPECT_EQ(hitsB, (long)(2 * T - 1));
  // prefetched == per-timestep, bit for bit
  EXPECT_EQ(A.values.size(), B.values.size());
  for (size_t i = 0; i < A.values.size(); ++i) {
    EXPECT_TRUE(A.values[i] == B.values[i]);
    EXPECT_EQ(A.convex[i]->hinges.size(), B.convex[i]->hinges.size());
    for (size_t h = 0; h < A.convex[i]->hinges.size() && h < B.convex[i]->hinges.size(); ++h) {
      const AffExpr &a = A.convex[i]->hinges[h].first, &b = B.convex[i]->hinges[h].first;
      EXPECT_TRUE(a.constant == b.constant && a.coeffs == b.coeffs);
    }
  }
  // the same hinges as the device assembles them (tohc_fetch_hinges on the slab the prefetch left) against
  // what sco would hand BPMPD for the reference's cost objects: ConvexObjective::addHinge (modeling.cpp:22-29: row
  // `viol - h <= 0`) -> BPMPDModel::optimize (bpmpd_interface.cpp:352-369: rhs = -constant; per variable the (row, value)
  // list through simplify2, solver_interface.cpp:23-37: exact zeros dropped, rows ascending).  Bit for bit.
  {
    tohc_hinges H;
    EXPECT_EQ(tohc_fetch_hinges(hip->GetEngine(), W.dist_pen, W.dist_pen, &H), (int)TOHC_OK);
    EXPECT_EQ((int)H.n_problems, 1);
    EXPECT_EQ((int)H.T, T);
    EXPECT_EQ((int)H.K, K);
    // rows of the device block are in canonical order: (t, discrete before cast); the costs come discrete t = 0.., then cast
    size_t nrows = 0;
    for (auto& c : B.convex) nrows += c->hinges.size();
    EXPECT_EQ((size_t)H.n_rows, nrows);
    vector<int> rowOfHinge;   // device row of the h-th hinge in cost order
    {
      vector<vector<int> > rowsOf(2 * T);   // [t] discrete, [T + t] cast
      for (int i = 0; i < (int)H.n_rows; ++i) rowsOf[(H.row_t[i] & TOHC_HINGE_CAST) ? T + (H.row_t[i] & ~TOHC_HINGE_CAST) : H.row_t[i]].push_back(i);
      for (size_t ci = 0; ci < B.convex.size(); ++ci) {
        EXPECT_EQ(rowsOf[ci].size(), B.convex[ci]->hinges.size());
        for (int r : rowsOf[ci]) rowOfHinge.push_back(r);
      }
    }
    if ((size_t)H.n_rows == nrows && rowOfHinge.size() == nrows) {
      std::map<int, std::map<int, double> > var2cnt;   // var -> (row -> value), what simplify2 leaves
      size_t h = 0;
      for (auto& c : B.convex)
        for (auto& hg : c->hinges) {
          const AffExpr& aff = hg.first;
          const int row = rowOfHinge[h++];
          EXPECT_TRUE(H.rhs[row] == -aff.constant);
          for (size_t j = 0; j < aff.coeffs.size(); ++j)
            if (aff.coeffs[j] != 0) var2cnt[aff.vars[j].var_index][row] += aff.coeffs[j];
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
  // CollisionConstraint is the same arithmetic scaled by coeff (collision_terms.cpp:244-262)
  {
    VarVector v(K);
    for (int k = 0; k < K; ++k) v[k] = Var((T / 2) * K + k);
    CollisionConstraint cnt(W.dist_pen, W.coeff, W.rad, v);
    DblVec val = cnt.value(x);
    ConvexConstraintsPtr cc = cnt.convex(x);
    EXPECT_EQ(val.size(), cc->ineqs.size());
    double sum = 0;
    for (size_t i = 0; i < val.size(); ++i) {
      sum += val[i];
      if (val[i] > 0) EXPECT_NEAR(cc->ineqs[i].value(x), val[i], 1e-9);   // the linearization is exact at x
    }
    EXPECT_NEAR(sum, A.values[T / 2], 1e-12);
  }

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
      dumpExpr(f, B.convex[i]->hinges[h].first);   // the hinge argument dist_pen - dist(x), as addHinge received it
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
    else { fprintf(stderr, "usage: host_unit box_distance | continuous_boxes | linearize IN OUT\n"); return 2; }
  } catch (const std::exception& e) {
    printf("EXCEPTION %s\n", e.what());
    return 3;
  }
  printf("%s: %s (%d failures)\n", cmd.c_str(), g_failures ? "FAILED" : "PASSED", g_failures);
  return g_failures ? 1 : 0;
}
