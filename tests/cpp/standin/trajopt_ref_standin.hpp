// trajopt_ref_standin.hpp -- TEST SCAFFOLDING, not part of the product.
//
// The HIP checker (trajopt_amd/host/hip_collision_checker.{hpp,cpp}) is written against the reference's own headers:
//   Collision, CastCollisionType, CollisionFilterGroups, CollisionChecker   src/trajopt/collision_checker.{hpp,cpp}
//   Configuration, RobotAndDOF                                              src/trajopt/configuration_space.{hpp,cpp}
// and is called by the reference's own collision terms (upstream: unchanged files; here tests/cpp/host_unit.cpp drives the
// checker the way they do).  Upstream keeps all of those files.  They need OpenRAVE, Boost and Eigen, none of which is in this image, so the
// tests compile the checker against THIS stand-in instead: the same declarations, name for name, over the OpenRAVE
// stand-in types of rave_shim.hpp and a sliver of sco (Var, AffExpr, addHinge, addIneqCnt) as plain data, and the same
// function bodies restated inline (each cites the reference lines it follows).  It earns no component credit and is
// linked into the test driver (tests/cpp/host_unit) only.
//
// The two additions a maintainer makes to the reference's headers are marked ADDITION below.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <iostream>
#include <set>
#include <unordered_map>
#include <utility>

#include "rave_shim.hpp"

namespace trajopt {

using std::vector;
using OR::KinBody;
typedef vector<double> DblVec;
typedef vector<int> IntVec;

// typedefs.hpp:29 -- row-major [T][K]
struct TrajArray {
  int m_rows, m_cols;
  DblVec m_data;
  TrajArray(int r = 0, int c = 0) : m_rows(r), m_cols(c), m_data((size_t)r * c, 0.0) {}
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }
  double& operator()(int r, int c) { return m_data[(size_t)r * m_cols + c]; }
  double operator()(int r, int c) const { return m_data[(size_t)r * m_cols + c]; }
  DblVec row(int r) const { return DblVec(m_data.begin() + (size_t)r * m_cols, m_data.begin() + (size_t)(r + 1) * m_cols); }
  const double* data() const { return m_data.data(); }
};

// ---- the sliver of sco the collision terms are written against -----------------------------------------------------
struct Var {
  int var_index;
  Var(int i = -1) : var_index(i) {}
};
typedef vector<Var> VarVector;
struct AffExpr {   // sco_common.hpp: constant + sum coeffs[i] * vars[i]
  double constant;
  DblVec coeffs;
  VarVector vars;
  explicit AffExpr(double c = 0) : constant(c) {}
  double value(const DblVec& x) const {
    double out = constant;
    for (size_t i = 0; i < coeffs.size(); ++i) out += coeffs[i] * x[vars[i].var_index];
    return out;
  }
};
inline void exprInc(AffExpr& a, const AffExpr& b) {
  a.constant += b.constant;
  a.coeffs.insert(a.coeffs.end(), b.coeffs.begin(), b.coeffs.end());
  a.vars.insert(a.vars.end(), b.vars.begin(), b.vars.end());
}
inline void exprInc(AffExpr& a, double b) { a.constant += b; }
inline void exprScale(AffExpr& a, double s) {
  a.constant *= s;
  for (double& c : a.coeffs) c *= s;
}
inline AffExpr exprSub(AffExpr a, const AffExpr& b) {
  a.constant -= b.constant;
  for (size_t i = 0; i < b.coeffs.size(); ++i) a.coeffs.push_back(-b.coeffs[i]), a.vars.push_back(b.vars[i]);
  return a;
}
inline AffExpr exprMult(AffExpr a, double s) { exprScale(a, s); return a; }
inline AffExpr varDot(const DblVec& g, const VarVector& vars) {
  AffExpr out;
  out.coeffs = g, out.vars = vars;
  return out;
}
inline DblVec getDblVec(const DblVec& x, const VarVector& vars) {
  DblVec out(vars.size());
  for (size_t i = 0; i < vars.size(); ++i) out[i] = x[vars[i].var_index];
  return out;
}
inline double pospart(double x) { return x > 0 ? x : 0; }
struct Model {};            // solver_interface.hpp: the QP under construction (opaque here)
struct ConvexObjective {   // modeling.hpp: only what CollisionCost::convex calls
  vector<std::pair<AffExpr, double> > hinges;
  explicit ConvexObjective(Model* = NULL) {}
  void addHinge(const AffExpr& e, double coeff) { hinges.push_back(std::make_pair(e, coeff)); }
};
struct ConvexConstraints {
  vector<AffExpr> ineqs;
  void addIneqCnt(const AffExpr& e) { ineqs.push_back(e); }
};
typedef std::shared_ptr<ConvexObjective> ConvexObjectivePtr;
typedef std::shared_ptr<ConvexConstraints> ConvexConstraintsPtr;
class Cost {   // modeling.hpp:94-110
 public:
  virtual double value(const vector<double>&) = 0;
  virtual ConvexObjectivePtr convex(const vector<double>& x, Model* model) = 0;
  virtual VarVector getVars() { return VarVector(); }
  std::string name() { return name_; }
  void setName(const std::string& name) { name_ = name; }
  Cost() : name_("unnamed") {}
  virtual ~Cost() {}

 protected:
  std::string name_;
};
typedef std::shared_ptr<Cost> CostPtr;

// ---- collision_checker.hpp ------------------------------------------------------------------------------------------
enum CastCollisionType { CCType_None, CCType_Time0, CCType_Time1, CCType_Between };

struct Collision {
  const KinBody::Link* linkA;
  const KinBody::Link* linkB;
  OR::Vector ptA, ptB, normalB2A;
  OR::Vector ptB1;
  double distance;
  float weight, time;
  CastCollisionType cctype;
  // the reference's constructor ignores its `time` argument (collision_checker.hpp:28); so does this one
  Collision(const KinBody::Link* linkA_, const KinBody::Link* linkB_, const OR::Vector& ptA_, const OR::Vector& ptB_,
            const OR::Vector& normalB2A_, double distance_, float weight_ = 1, float time_ = 0)
      : linkA(linkA_), linkB(linkB_), ptA(ptA_), ptB(ptB_), normalB2A(normalB2A_), distance(distance_), weight(weight_),
        time(0), cctype(CCType_None), childA(0), childB(0), const0(0), const1(0) { (void)time_; }
  // ---- additions of the HIP checker (not in the reference) ----
  int childA, childB;     // geometry index inside linkA / linkB
  double const0, const1;  // dist(x) ~ const0 + grad0 . x_t ; cast: (1-time)(const0 + grad0 . x_t) + time (const1 + grad1 . x_t+1)
  DblVec grad0, grad1;    // empty when the query had no kinematic model (LinksVsAll on explicit link lists)
};
std::ostream& operator<<(std::ostream&, const Collision&);

enum CollisionFilterGroups { RobotFilter = 1, KinBodyFilter = 2 };

class Configuration;

class CollisionChecker : public OR::UserData {
 public:
  virtual void AllVsAll(vector<Collision>& collisions) = 0;
  virtual void LinkVsAll(const KinBody::Link& link, vector<Collision>& collisions, short filterMask) = 0;
  virtual void LinksVsAll(const vector<KinBody::LinkPtr>& links, vector<Collision>& collisions, short filterMask) = 0;
  void BodyVsAll(const KinBody& body, vector<Collision>& collisions, short filterMask = -1) { LinksVsAll(body.GetLinks(), collisions, filterMask); }
  virtual void SetContactDistance(float distance) = 0;
  virtual double GetContactDistance() = 0;
  virtual void ContinuousCheckTrajectory(const TrajArray&, Configuration&, vector<Collision>&) { throw std::runtime_error("not implemented"); }
  virtual void CastVsAll(Configuration&, const vector<KinBody::LinkPtr>&, const DblVec&, const DblVec&, vector<Collision>&) { throw std::runtime_error("not implemented"); }
  void IgnoreZeroStateSelfCollisions();
  void IgnoreZeroStateSelfCollisions(OR::KinBodyPtr body);
  virtual void ExcludeCollisionPair(const KinBody::Link& link0, const KinBody::Link& link1) = 0;
  virtual void IncludeCollisionPair(const KinBody::Link& link0, const KinBody::Link& link1) = 0;
  virtual bool RayCastCollision(const OR::Vector& point1, const OR::Vector& point2) = 0;
  // ADDITION (empty default, so the Bullet checker is unaffected): the configuration LinksVsAll linearizes against; the
  // evaluators call it before they set the DOF values
  virtual void BindConfiguration(Configuration&) {}
  OR::EnvironmentBaseConstPtr GetEnv() { return m_env; }
  virtual ~CollisionChecker() {}
  static std::shared_ptr<CollisionChecker> GetOrCreate(OR::EnvironmentBase& env);

 protected:
  CollisionChecker(OR::EnvironmentBaseConstPtr env) : m_env(env) {}
  OR::EnvironmentBaseConstPtr m_env;
};
typedef std::shared_ptr<CollisionChecker> CollisionCheckerPtr;
CollisionCheckerPtr CreateCollisionChecker(OR::EnvironmentBaseConstPtr env);

// ---- configuration_space.hpp ----------------------------------------------------------------------------------------
class Configuration {
 public:
  virtual void SetDOFValues(const DblVec& dofs) = 0;
  virtual DblVec GetDOFValues() = 0;
  virtual int GetDOF() const = 0;
  virtual OR::EnvironmentBasePtr GetEnv() = 0;
  virtual bool DoesAffect(const KinBody::Link& link) = 0;
  virtual void GetAffectedLinks(vector<KinBody::LinkPtr>& links, bool only_with_geom, vector<int>& link_inds) = 0;
  virtual ~Configuration() {}
};
typedef std::shared_ptr<Configuration> ConfigurationPtr;

class RobotAndDOF : public Configuration {
 public:
  RobotAndDOF(OR::RobotBasePtr robot_, const IntVec& joint_inds_, int affinedofs_ = 0, const OR::Vector& rotationaxis_ = OR::Vector(0, 0, 1),
              OR::EnvironmentBasePtr env = OR::EnvironmentBasePtr())
      : robot(robot_), joint_inds(joint_inds_), affinedofs(affinedofs_), rotationaxis(rotationaxis_), m_env(env) {}
  void SetDOFValues(const DblVec& dofs);
  DblVec GetDOFValues();
  int GetDOF() const;
  OR::EnvironmentBasePtr GetEnv() { return m_env; }
  bool DoesAffect(const KinBody::Link& link);
  void GetAffectedLinks(vector<KinBody::LinkPtr>& links, bool only_with_geom, vector<int>& link_inds);
  OR::RobotBasePtr GetRobot() const { return robot; }
  IntVec GetJointIndices() const { return joint_inds; }

  OR::RobotBasePtr robot;
  IntVec joint_inds;
  int affinedofs;
  OR::Vector rotationaxis;

 private:
  OR::EnvironmentBasePtr m_env;
};
typedef std::shared_ptr<RobotAndDOF> RobotAndDOFPtr;

// ===================================================================================================================
// function bodies (collision_checker.cpp, configuration_space.cpp), restated inline
// ===================================================================================================================
using OR::RobotBase;
using OR::Transform;
using OR::Vector;
inline size_t hashRange(const DblVec& x, size_t seed = 0) {   // boost::hash_range (collision_terms.cpp:82-84)
  std::hash<double> h;
  for (double v : x) seed ^= h(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
  return seed;
}
inline int affineCount(int a) { return ((a & 1) != 0) + ((a & 2) != 0) + ((a & 4) != 0) + ((a & 8) != 0); }

inline std::ostream& operator<<(std::ostream& o, const Collision& c) {   // collision_checker.cpp:68-77
  auto v = [&](const Vector& a) { o << "[" << a.x << ", " << a.y << ", " << a.z << "]"; };
  o << (c.linkA ? c.linkA->GetName() : "NULL") << "--" << (c.linkB ? c.linkB->GetName() : "NULL") << " distance: " << c.distance << " normal: ";
  v(c.normalB2A); o << " ptA: "; v(c.ptA); o << " ptB: "; v(c.ptB);
  o << " time: " << c.time << " weight: " << c.weight;
  return o;
}

// ===================================================================================================================
// CollisionChecker (collision_checker.cpp)
// ===================================================================================================================
inline std::shared_ptr<CollisionChecker> CollisionChecker::GetOrCreate(OR::EnvironmentBase& env) {   // :16-27
  OR::UserDataPtr ud = env.m_userdata["trajopt_cc"];
  if (!ud) {
    ud = CreateCollisionChecker(env.shared_from_this());
    env.m_userdata["trajopt_cc"] = ud;
  }
  return std::dynamic_pointer_cast<CollisionChecker>(ud);
}

inline void CollisionChecker::IgnoreZeroStateSelfCollisions(OR::KinBodyPtr body) {   // :41-56
  // KinBodyStateSaver
  Transform savedT = body->GetTransform();
  RobotBase* rb = dynamic_cast<RobotBase*>(body.get());
  DblVec savedQ;
  if (rb) savedQ = rb->GetDOFValues(), rb->SetDOFValues(DblVec(rb->GetDOF(), 0.0));
  body->SetTransform(Transform(Vector(1, 0, 0, 0), Vector(0, 0, 10)));
  vector<Collision> collisions;
  BodyVsAll(*body, collisions);
  for (size_t i = 0; i < collisions.size(); ++i) ExcludeCollisionPair(*collisions[i].linkA, *collisions[i].linkB);
  if (rb) rb->SetDOFValues(savedQ);
  body->SetTransform(savedT);
}

inline void CollisionChecker::IgnoreZeroStateSelfCollisions() {   // :58-66
  vector<OR::KinBodyPtr> bodies;
  GetEnv()->GetBodies(bodies);
  for (auto& b : bodies) IgnoreZeroStateSelfCollisions(b);
}

// ===================================================================================================================
// RobotAndDOF (configuration_space.cpp)
// ===================================================================================================================
inline int RobotAndDOF::GetDOF() const { return (int)joint_inds.size() + affineCount(affinedofs); }   // :80-82

inline void RobotAndDOF::SetDOFValues(const DblVec& dofs) {   // :12-23
  if ((int)dofs.size() != GetDOF()) throw std::runtime_error("RobotAndDOF::SetDOFValues: wrong number of values");
  if (affinedofs != 0) {
    // RaveGetTransformFromAffineDOFValues on a default (identity) transform: X, Y, Z, then the angle about rotationaxis
    Transform T;
    size_t i = joint_inds.size();
    if (affinedofs & OR::DOF_X) T.trans.x = dofs[i++];
    if (affinedofs & OR::DOF_Y) T.trans.y = dofs[i++];
    if (affinedofs & OR::DOF_Z) T.trans.z = dofs[i++];
    if (affinedofs & OR::DOF_RotationAxis) {
      double a = dofs[i++], s = sin(a / 2);
      T.rot = Vector(cos(a / 2), rotationaxis.x * s, rotationaxis.y * s, rotationaxis.z * s);
    }
    robot->SetTransform(T);
    if (joint_inds.size() > 0) robot->SetDOFValues(DblVec(dofs.begin(), dofs.begin() + joint_inds.size()), joint_inds);
  } else {
    robot->SetDOFValues(dofs, joint_inds);
  }
}

inline DblVec RobotAndDOF::GetDOFValues() {   // :25-35
  DblVec out;
  for (int j : joint_inds) out.push_back(robot->GetDOFValues().at(j));
  if (affinedofs != 0) {   // RaveGetAffineDOFValuesFromTransform: an angle set through SetDOFValues comes back to ~1e-16, not bit for bit
    const Transform& T = robot->GetTransform();
    if (affinedofs & OR::DOF_X) out.push_back(T.trans.x);
    if (affinedofs & OR::DOF_Y) out.push_back(T.trans.y);
    if (affinedofs & OR::DOF_Z) out.push_back(T.trans.z);
    if (affinedofs & OR::DOF_RotationAxis)
      out.push_back(2 * atan2(T.rot.y * rotationaxis.x + T.rot.z * rotationaxis.y + T.rot.w * rotationaxis.z, T.rot.x));
  }
  return out;
}

inline bool RobotAndDOF::DoesAffect(const KinBody::Link& link) {   // :97-101, rave_utils.cpp:69-74
  if (affinedofs > 0) return true;
  if (link.GetParent() != robot.get()) return false;
  const vector<RobotBase::Joint>& J = robot->GetJoints();
  for (int i = link.GetIndex(); i >= 0; i = J[i].parent)
    if (J[i].type != OR::JointFixed && std::find(joint_inds.begin(), joint_inds.end(), J[i].dof) != joint_inds.end()) return true;
  return false;
}

inline void RobotAndDOF::GetAffectedLinks(vector<KinBody::LinkPtr>& links, bool only_with_geom, vector<int>& link_inds) {   // :110-118
  links.clear(), link_inds.clear();
  for (const KinBody::LinkPtr& link : robot->GetLinks())
    if (DoesAffect(*link) && !(only_with_geom && link->GetGeometries().empty())) links.push_back(link), link_inds.push_back(link->GetIndex());
  vector<OR::KinBodyPtr> grabbed;   // :120-131
  robot->GetGrabbed(grabbed);
  for (const OR::KinBodyPtr& body : grabbed) {
    KinBody::LinkPtr grabberLink = robot->IsGrabbing(body);
    for (const KinBody::LinkPtr& link : body->GetLinks())
      if (link->GetGeometries().size() > 0) links.push_back(link), link_inds.push_back(grabberLink->GetIndex());
  }
}

}  // namespace trajopt
