// collision_terms_standin.hpp -- TEST SCAFFOLDING, not part of the product: the CALLER of the checker.
//
//   Cache                                                                   src/trajopt/cache.hxx:3-24
//   CollisionEvaluator, SingleTimestep/CastCollisionEvaluator, CollisionsToDistances,
//   CollisionsToDistanceExpressions, CollisionCost, CollisionConstraint     src/trajopt/collision_terms.{hpp,cpp}
// Upstream these are the reference's own files; they need sco (Eigen, Boost) and OpenRAVE, absent from this image, so the
// test driver (tests/cpp/host_unit.cpp) compiles this restatement over the stand-in types instead.  Two differences from
// the reference, both marked: CollisionsToDistanceExpressions assembles the AffExpr from the linearization the checker
// put into the Collision (no PositionJacobian call), and the evaluators tell the checker which configuration they use.
#pragma once
#include "trajopt_ref_standin.hpp"

namespace trajopt {

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


// ===================================================================================================================
// collision_terms.cpp
// ===================================================================================================================
inline void CollisionsToDistances(const vector<Collision>& collisions, const Link2Int& link2ind, DblVec& dists) {   // :22-35
  dists.clear();
  dists.reserve(collisions.size());
  for (const Collision& col : collisions)
    if (link2ind.find(col.linkA) != link2ind.end() || link2ind.find(col.linkB) != link2ind.end()) dists.push_back(col.distance);
}

// :37-62.  The reference computes, per robot-side link, dist_grad = +-n^T J(link, pt) through PositionJacobian and adds
// varDot(dist_grad, vars) - dist_grad . dofvals.  The kernels did exactly that sum over both sides in k_finish:
// grad = sum of the sides' rows, const = distance - grad . dofvals.  So the expression is assembled, not recomputed.
inline void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration&, const Link2Int& link2ind, const VarVector& vars,
                                     const DblVec&, vector<AffExpr>& exprs, bool isTimestep1) {
  exprs.clear();
  exprs.reserve(collisions.size());
  for (const Collision& col : collisions) {
    if (link2ind.find(col.linkA) == link2ind.end() && link2ind.find(col.linkB) == link2ind.end()) continue;
    const DblVec& g = isTimestep1 ? col.grad1 : col.grad0;
    if (g.size() != vars.size())
      throw std::runtime_error("CollisionsToDistanceExpressions: this contact carries no linearization (it came from a query on an explicit "
                               "link list; PositionJacobian needs OpenRAVE)");
    AffExpr dist(isTimestep1 ? col.const1 : col.const0);
    exprInc(dist, varDot(g, vars));
    exprs.push_back(dist);
  }
}

inline void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration& rad, const Link2Int& link2ind, const VarVector& vars0,
                                     const VarVector& vars1, const DblVec& vals0, const DblVec& vals1, vector<AffExpr>& exprs) {   // :64-80
  vector<AffExpr> exprs0, exprs1;
  CollisionsToDistanceExpressions(collisions, rad, link2ind, vars0, vals0, exprs0, false);
  CollisionsToDistanceExpressions(collisions, rad, link2ind, vars1, vals1, exprs1, true);
  exprs.resize(exprs0.size());
  for (size_t i = 0; i < exprs0.size(); ++i) {
    exprScale(exprs0[i], (1 - collisions[i].time));
    exprScale(exprs1[i], collisions[i].time);
    exprs[i] = AffExpr(0);
    exprInc(exprs[i], exprs0[i]);
    exprInc(exprs[i], exprs1[i]);
    // cleanupAff's result is discarded in the reference (:78): tiny coefficients are kept
  }
}

inline void CollisionEvaluator::GetCollisionsCached(const DblVec& x, vector<Collision>& collisions) {   // :86-98
  size_t key = hashRange(getDblVec(x, GetVars()));
  vector<Collision>* it = m_cache.get(key);
  if (it != NULL) {
    collisions = *it;
  } else {
    CalcCollisions(x, collisions);
    m_cache.put(key, collisions);
  }
}

inline SingleTimestepCollisionEvaluator::SingleTimestepCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars)   // :100-115
    : m_env(rad->GetEnv()), m_cc(CollisionChecker::GetOrCreate(*m_env)), m_rad(rad), m_vars(vars), m_filterMask(-1) {
  vector<int> inds;
  rad->GetAffectedLinks(m_links, true, inds);
  for (size_t i = 0; i < m_links.size(); ++i) m_link2ind[m_links[i].get()] = inds[i];
}

inline void SingleTimestepCollisionEvaluator::CalcCollisions(const DblVec& x, vector<Collision>& collisions) {   // :118-122
  DblVec dofvals = getDblVec(x, m_vars);
  m_cc->BindConfiguration(*m_rad);   // ADDITION: the one line the reference's evaluators do not have
  m_rad->SetDOFValues(dofvals);
  m_cc->LinksVsAll(m_links, collisions, m_filterMask);
}

inline void SingleTimestepCollisionEvaluator::CalcDists(const DblVec& x, DblVec& dists) {   // :124-128
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  CollisionsToDistances(collisions, m_link2ind, dists);
}

inline void SingleTimestepCollisionEvaluator::CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs) {   // :131-136
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  DblVec dofvals = getDblVec(x, m_vars);
  CollisionsToDistanceExpressions(collisions, *m_rad, m_link2ind, m_vars, dofvals, exprs, false);
}

inline CastCollisionEvaluator::CastCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :140-154
    : m_env(rad->GetEnv()), m_cc(CollisionChecker::GetOrCreate(*m_env)), m_rad(rad), m_vars0(vars0), m_vars1(vars1), m_filterMask(-1) {
  vector<int> inds;
  rad->GetAffectedLinks(m_links, true, inds);
  for (size_t i = 0; i < m_links.size(); ++i) m_link2ind[m_links[i].get()] = inds[i];
}

inline void CastCollisionEvaluator::CalcCollisions(const DblVec& x, vector<Collision>& collisions) {   // :156-161
  DblVec dofvals0 = getDblVec(x, m_vars0);
  DblVec dofvals1 = getDblVec(x, m_vars1);
  m_rad->SetDOFValues(dofvals0);
  m_cc->CastVsAll(*m_rad, m_links, dofvals0, dofvals1, collisions);
}

inline void CastCollisionEvaluator::CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs) {   // :162-168
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  DblVec dofvals0 = getDblVec(x, m_vars0);
  DblVec dofvals1 = getDblVec(x, m_vars1);
  CollisionsToDistanceExpressions(collisions, *m_rad, m_link2ind, m_vars0, m_vars1, dofvals0, dofvals1, exprs);
}

inline void CastCollisionEvaluator::CalcDists(const DblVec& x, DblVec& dists) {   // :169-173
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  CollisionsToDistances(collisions, m_link2ind, dists);
}

inline CollisionCost::CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars)   // :196-199
    : m_calc(new SingleTimestepCollisionEvaluator(rad, vars)), m_dist_pen(dist_pen), m_coeff(coeff) {}
inline CollisionCost::CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :201-204
    : m_calc(new CastCollisionEvaluator(rad, vars0, vars1)), m_dist_pen(dist_pen), m_coeff(coeff) {}

inline ConvexObjectivePtr CollisionCost::convex(const DblVec& x) {   // :205-214
  ConvexObjectivePtr out(new ConvexObjective);
  vector<AffExpr> exprs;
  m_calc->CalcDistExpressions(x, exprs);
  for (size_t i = 0; i < exprs.size(); ++i) {
    AffExpr viol = exprSub(AffExpr(m_dist_pen), exprs[i]);
    out->addHinge(viol, m_coeff);
  }
  return out;
}

inline double CollisionCost::value(const DblVec& x) {   // :215-223
  DblVec dists;
  m_calc->CalcDists(x, dists);
  double out = 0;
  for (size_t i = 0; i < dists.size(); ++i) out += pospart(m_dist_pen - dists[i]) * m_coeff;
  return out;
}

inline CollisionConstraint::CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars)   // :233-237
    : m_calc(new SingleTimestepCollisionEvaluator(rad, vars)), m_dist_pen(dist_pen), m_coeff(coeff) {}
inline CollisionConstraint::CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :239-243
    : m_calc(new CastCollisionEvaluator(rad, vars0, vars1)), m_dist_pen(dist_pen), m_coeff(coeff) {}

inline ConvexConstraintsPtr CollisionConstraint::convex(const DblVec& x) {   // :244-253
  ConvexConstraintsPtr out(new ConvexConstraints);
  vector<AffExpr> exprs;
  m_calc->CalcDistExpressions(x, exprs);
  for (size_t i = 0; i < exprs.size(); ++i) {
    AffExpr viol = exprSub(AffExpr(m_dist_pen), exprs[i]);
    out->addIneqCnt(exprMult(viol, m_coeff));
  }
  return out;
}

inline DblVec CollisionConstraint::value(const DblVec& x) {   // :254-262
  DblVec dists;
  m_calc->CalcDists(x, dists);
  DblVec out(dists.size());
  for (size_t i = 0; i < dists.size(); ++i) out[i] = pospart(m_dist_pen - dists[i]) * m_coeff;
  return out;
}


}  // namespace trajopt
