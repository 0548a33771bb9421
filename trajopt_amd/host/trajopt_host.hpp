// trajopt_host.hpp -- C++ host side of the MI355X collision path, above the C-ABI (include/trajopt_hipcc.h).
//
// Mirrors, name for name, the reference's interface for this path so that the engine is a drop-in:
//   Collision, CastCollisionType, CollisionFilterGroups, CollisionChecker   src/trajopt/collision_checker.hpp:12-87
//   Configuration, RobotAndDOF                                              src/trajopt/configuration_space.{hpp,cpp}
//   Cache                                                                   src/trajopt/cache.hxx:3-24
//   CollisionEvaluator, SingleTimestep/CastCollisionEvaluator, CollisionsToDistances,
//   CollisionsToDistanceExpressions, CollisionCost, CollisionConstraint     src/trajopt/collision_terms.{hpp,cpp}
//   HIPCollisionChecker = what BulletCollisionChecker is in the reference   src/trajopt/bullet_collision_checker.cpp:321-1039
// sco is out of scope; the few sco types the collision terms are written against (Var, AffExpr,
// ConvexObjective::addHinge, ConvexConstraints::addIneqCnt; src/sco/{sco_common,modeling}.hpp) are declared here as
// plain data so the terms compile and their output can be checked.  OpenRAVE types come from rave_shim.hpp.
//
// What is new relative to the reference: HIPCollisionChecker::PrefetchTrajectory sends all timesteps
// of a trajectory to the GPU in one call; the unchanged per-timestep LinksVsAll / CastVsAll then answer from that
// slab.  Collision carries the fused linearization (const0/grad0, const1/grad1) the kernels computed, so
// CollisionsToDistanceExpressions needs no PositionJacobian call (configuration_space.cpp:83-89).
#pragma once
#include <cstring>
#include <iostream>
#include <set>
#include <unordered_map>
#include <utility>

#include "rave_shim.hpp"
#include "trajopt_hipcc.h"

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
struct ConvexObjective {   // modeling.hpp: only what CollisionCost::convex calls
  vector<std::pair<AffExpr, double> > hinges;
  void addHinge(const AffExpr& e, double coeff) { hinges.push_back(std::make_pair(e, coeff)); }
};
struct ConvexConstraints {
  vector<AffExpr> ineqs;
  void addIneqCnt(const AffExpr& e) { ineqs.push_back(e); }
};
typedef std::shared_ptr<ConvexObjective> ConvexObjectivePtr;
typedef std::shared_ptr<ConvexConstraints> ConvexConstraintsPtr;

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

// ---- the MI355X checker ------------------------------------------------------------------------------------------------
class HIPCollisionChecker : public CollisionChecker {
 public:
  explicit HIPCollisionChecker(OR::EnvironmentBaseConstPtr env);
  ~HIPCollisionChecker();

  void SetContactDistance(float distance);
  double GetContactDistance() { return m_contactDistance; }
  void AllVsAll(vector<Collision>& collisions);
  void LinksVsAll(const vector<KinBody::LinkPtr>& links, vector<Collision>& collisions, short filterMask);
  void LinkVsAll(const KinBody::Link& link, vector<Collision>& collisions, short filterMask);
  void ContinuousCheckTrajectory(const TrajArray& traj, Configuration& rad, vector<Collision>& collisions);
  void CastVsAll(Configuration& rad, const vector<KinBody::LinkPtr>& links, const DblVec& startjoints, const DblVec& endjoints,
                 vector<Collision>& collisions);
  void ExcludeCollisionPair(const KinBody::Link& link0, const KinBody::Link& link1);
  void IncludeCollisionPair(const KinBody::Link& link0, const KinBody::Link& link1);
  bool RayCastCollision(const OR::Vector& point1, const OR::Vector& point2);
  /** Batched form (no counterpart in the reference): closest hit fraction per ray, 1 = nothing hit. points: xyz triples. */
  vector<float> RayCastCollisions(const vector<double>& points1, const vector<double>& points2);

  // ---- additions (no counterpart in the reference) ----
  /** The active configuration LinksVsAll linearizes against (the evaluators call this from their constructors; CastVsAll
      and ContinuousCheckTrajectory bind the one they are handed). */
  void BindConfiguration(Configuration& rad);
  /** One GPU call for a whole trajectory: the discrete contacts of every timestep (LinksVsAll, filterMask) and / or
      the cast contacts of every (t, t+1) pair, with their linearizations.  Later LinksVsAll / CastVsAll calls at those
      DOF vectors are answered from this slab.  Invalidated by any change of the world, d, or the exclusions. */
  void PrefetchTrajectory(Configuration& rad, const TrajArray& traj, bool discrete, bool cast, short filterMask = -1);
  /** Robot link poses come from the engine's FK kernel (OpenRAVE's FK in the reference). */
  void UpdateRobotLinkTransforms();
  tohc_engine* GetEngine() { return m_e; }
  long NumPrefetchHits() const { return m_prefetchHits; }
  long NumEngineCalls() const { return m_engineCalls; }

 private:
  typedef std::pair<const KinBody::Link*, const KinBody::Link*> LinkPair;
  struct Prefetched {
    DblVec q0, q1;   // q1 empty: discrete
    vector<Collision> collisions;
  };
  void chk(int rc) const;
  void UpdateFromRave();            // UpdateBulletFromRave, bullet_collision_checker.cpp:629-655
  void Rebuild(const vector<OR::KinBodyPtr>& bodies);   // AddKinBody / CollisionObjectFromLink / createShapePrimitive, :134-255,535-561
  void UpdateAllowedCollisionMatrix();                  // :616-627
  void UploadRobot(RobotAndDOF& rad);
  bool RobotUploadIsCurrent(RobotAndDOF& rad) const;
  vector<double> CurrentPoses();    // [n_objects][7]
  Collision ToCollision(const tohc_record& r, const double* grads, int K) const;
  bool IsAffectedSet(const vector<KinBody::LinkPtr>& links) const;
  const Prefetched* FindPrefetched(const DblVec& q0, const DblVec* q1, short filterMask) const;
  void InvalidatePrefetch() { m_prefetch.clear(), m_prefetchIndex.clear(); }

  tohc_engine* m_e;
  double m_contactDistance;
  vector<OR::KinBodyPtr> m_prevbodies;
  vector<const KinBody::Link*> m_obj2link;
  std::map<const KinBody::Link*, int> m_link2obj;
  vector<uint8_t> m_allowed;        // m_allowedCollisionMatrix, :332
  vector<double> m_uploadedT;       // poses the scene upload carries, [n_objects][7]
  std::set<LinkPair> m_excludedPairs;
  // robot binding
  RobotAndDOF* m_bound;
  std::shared_ptr<RobotAndDOF> m_boundKeep;
  OR::RobotBase* m_fkRobot;
  OR::RobotBase* m_upRobot;
  IntVec m_upJointInds;
  int m_upAffine;
  OR::Vector m_upRotAxis;
  DblVec m_upQ;
  OR::Transform m_upBase;
  vector<int> m_affectedObjs;       // object ids of the bound configuration's affected links with geometry
  unsigned long m_fkStamp;
  // prefetch
  vector<Prefetched> m_prefetch;
  std::unordered_multimap<size_t, int> m_prefetchIndex;
  short m_prefetchMask;
  long m_prefetchHits, m_engineCalls;
};

// ---- cache.hxx ---------------------------------------------------------------------------------------------------------
template <class KeyT, class ValueT, unsigned bufsize>
class Cache {
 public:
  KeyT keybuf[bufsize];   // circular buffer
  ValueT valbuf[bufsize];
  int m_i;
  Cache() : m_i(0) { memset(keybuf, 666, sizeof(keybuf)); }
  void put(const KeyT& key, const ValueT& value) {
    keybuf[m_i] = key, valbuf[m_i] = value;
    if (++m_i == (int)bufsize) m_i = 0;
  }
  ValueT* get(const KeyT& key) {
    for (unsigned i = 0; i < bufsize; ++i)
      if (keybuf[i] == key) return &valbuf[i];
    return NULL;
  }
};

// ---- collision_terms.hpp -----------------------------------------------------------------------------------------------
typedef std::map<const KinBody::Link*, int> Link2Int;

void CollisionsToDistances(const vector<Collision>& collisions, const Link2Int& link2ind, DblVec& dists);
void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration& rad, const Link2Int& link2ind, const VarVector& vars,
                                     const DblVec& dofvals, vector<AffExpr>& exprs, bool isTimestep1);
void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration& rad, const Link2Int& link2ind, const VarVector& vars0,
                                     const VarVector& vars1, const DblVec& vals0, const DblVec& vals1, vector<AffExpr>& exprs);

struct CollisionEvaluator {
  virtual void CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs) = 0;
  virtual void CalcDists(const DblVec& x, DblVec& exprs) = 0;
  virtual void CalcCollisions(const DblVec& x, vector<Collision>& collisions) = 0;
  void GetCollisionsCached(const DblVec& x, vector<Collision>&);
  virtual ~CollisionEvaluator() {}
  virtual VarVector GetVars() = 0;
  Cache<size_t, vector<Collision>, 3> m_cache;
};
typedef std::shared_ptr<CollisionEvaluator> CollisionEvaluatorPtr;

struct SingleTimestepCollisionEvaluator : public CollisionEvaluator {
  SingleTimestepCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars);
  void CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs);
  void CalcDists(const DblVec& x, DblVec& exprs);
  void CalcCollisions(const DblVec& x, vector<Collision>& collisions);
  VarVector GetVars() { return m_vars; }
  OR::EnvironmentBasePtr m_env;
  CollisionCheckerPtr m_cc;
  ConfigurationPtr m_rad;
  VarVector m_vars;
  Link2Int m_link2ind;
  vector<KinBody::LinkPtr> m_links;
  short m_filterMask;
};

struct CastCollisionEvaluator : public CollisionEvaluator {
  CastCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1);
  void CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs);
  void CalcDists(const DblVec& x, DblVec& exprs);
  void CalcCollisions(const DblVec& x, vector<Collision>& collisions);
  VarVector GetVars() {
    VarVector out = m_vars0;
    out.insert(out.end(), m_vars1.begin(), m_vars1.end());
    return out;
  }
  OR::EnvironmentBasePtr m_env;
  CollisionCheckerPtr m_cc;
  ConfigurationPtr m_rad;
  VarVector m_vars0, m_vars1;
  Link2Int m_link2ind;
  vector<KinBody::LinkPtr> m_links;
  short m_filterMask;
};

class CollisionCost {   // : public sco::Cost
 public:
  CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars);
  CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1);
  ConvexObjectivePtr convex(const DblVec& x);
  double value(const DblVec& x);
  VarVector getVars() { return m_calc->GetVars(); }
  CollisionEvaluatorPtr GetEvaluator() { return m_calc; }

 private:
  CollisionEvaluatorPtr m_calc;
  double m_dist_pen, m_coeff;
};

class CollisionConstraint {   // : public sco::IneqConstraint
 public:
  CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars);
  CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1);
  ConvexConstraintsPtr convex(const DblVec& x);
  DblVec value(const DblVec& x);
  VarVector getVars() { return m_calc->GetVars(); }

 private:
  CollisionEvaluatorPtr m_calc;
  double m_dist_pen, m_coeff;
};

}  // namespace trajopt
