// collision_prefetch_cost.hpp -- the hook that turns the reference's per-timestep collision terms into ONE GPU call per
// SQP evaluation, without touching sco or collision_terms.
//
// BasicTrustRegionSQP walks prob.getCosts() in order, handing every cost the WHOLE solution vector x: value(x) in
// evaluateCosts, convex(x, model) in convexifyCosts (src/sco/optimizers.cpp:36-56).  A CollisionPrefetchCost placed
// FIRST in that list is therefore called before any CollisionCost of the same pass: it reads the trajectory out of x,
// sends all of it to the GPU (HIPCollisionChecker::PrefetchTrajectory: discrete contacts of every timestep and / or cast
// contacts of every pair, with their linearizations, one launch sequence, one fetch) and contributes nothing itself --
// value 0, an empty ConvexObjective.  The unchanged per-timestep evaluators then call LinksVsAll / CastVsAll as they
// always did (collision_terms.cpp:118-122,156-161) and are answered from that slab.  The same x again (convexify right
// after the evaluate that accepted x) costs no second GPU call while the slab is still valid.
//
// Set-up, where the reference hatches its collision costs (problem_description.cpp:516-549):
//     prob.getCosts().insert(prob.getCosts().begin(), CostPtr(new CollisionPrefetchCost(prob.GetRAD(), rows, discrete, continuous)));
// with rows[t] = prob.GetVarRow(t).  With the Bullet checker (TRAJOPT_COLLISION_CHECKER=bullet) the cost does nothing.
#pragma once
#include "hip_collision_checker.hpp"   // upstream also: "sco/modeling.hpp" (Cost, ConvexObjective, Var)

namespace trajopt {

class CollisionPrefetchCost : public Cost {
 public:
  CollisionPrefetchCost(ConfigurationPtr rad, const vector<VarVector>& rows, bool discrete, bool cast, short filterMask = -1)
      : m_rad(rad), m_rows(rows), m_discrete(discrete), m_cast(cast), m_filterMask(filterMask), m_calls(0) {
    setName("collision_prefetch");
  }
  double value(const vector<double>& x) {
    Prefetch(x);
    return 0;
  }
  ConvexObjectivePtr convex(const vector<double>& x, Model* model) {
    Prefetch(x);
    return ConvexObjectivePtr(new ConvexObjective(model));   // no terms: the QP is unchanged
  }
  VarVector getVars() {
    VarVector out;
    for (const VarVector& r : m_rows) out.insert(out.end(), r.begin(), r.end());
    return out;
  }
  long NumPrefetches() const { return m_calls; }

 private:
  void Prefetch(const vector<double>& x) {
    HIPCollisionChecker* hip = dynamic_cast<HIPCollisionChecker*>(CollisionChecker::GetOrCreate(*m_rad->GetEnv()).get());
    if (!hip || m_rows.empty()) return;
    const int T = (int)m_rows.size(), K = (int)m_rows[0].size();
    TrajArray traj(T, K);
    DblVec flat((size_t)T * K);
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < K; ++k) traj(t, k) = flat[(size_t)t * K + k] = x[m_rows[t][k].var_index];
    if (hip->HasPrefetch() && flat == m_last) return;   // this trajectory is what the checker holds
    hip->PrefetchTrajectory(*m_rad, traj, m_discrete, m_cast, m_filterMask);
    m_last.swap(flat);
    ++m_calls;
  }
  ConfigurationPtr m_rad;
  vector<VarVector> m_rows;
  bool m_discrete, m_cast;
  short m_filterMask;
  DblVec m_last;
  long m_calls;
};

}  // namespace trajopt
