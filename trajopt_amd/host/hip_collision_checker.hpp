// hip_collision_checker.hpp -- C++ host side of the MI355X collision path, above the C-ABI (include/trajopt_hipcc.h):
// HIPCollisionChecker, the drop-in for the reference's BulletCollisionChecker
// (src/trajopt/bullet_collision_checker.cpp:321-1039) behind trajopt::CollisionChecker
// (src/trajopt/collision_checker.hpp:40-84), and the factory CreateCollisionChecker (:87).
//
// It is written against the reference's own headers (collision_checker.hpp, configuration_space.hpp).  Upstream this
// file includes those; in this image they cannot be compiled (OpenRAVE, Boost, Eigen are absent), so the build passes
// -Itests/cpp/standin and the include below finds the test stand-in of the same declarations.  Nothing of the stand-in
// is part of libtrajopt_host.so beyond what its inline functions instantiate here.
//
// What is new relative to the reference: PrefetchTrajectory sends all timesteps of a trajectory to the
// GPU in one call; the unchanged per-timestep LinksVsAll / CastVsAll then answer from that slab.  Collision carries the
// fused linearization (const0/grad0, const1/grad1) the kernels computed, so CollisionsToDistanceExpressions needs no
// PositionJacobian call (configuration_space.cpp:83-89).
#pragma once
#include <map>
#include <set>
#include <unordered_map>

#include "trajopt_ref_standin.hpp"   // upstream: "trajopt/collision_checker.hpp", "trajopt/configuration_space.hpp"
#include "trajopt_hipcc.h"

namespace trajopt {

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
  tohc_engine* GetEngine() { return m_e; }
  bool HasPrefetch() const { return !m_prefetch.empty(); }   // false again after anything invalidated it (world, d, exclusions)
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
  void CheckDeviceFK(RobotAndDOF& rad);                 // the FK guard: device FK of the uploaded tree == link->GetTransform()
  bool StateIsRepresentable(RobotAndDOF& rad) const;    // the DOF vector describes the robot's state (affine DOFs vs body transform)
  vector<double> GrabSignature(const OR::RobotBase& r) const;
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
  OR::RobotBase* m_upRobot;
  IntVec m_upJointInds;
  int m_upAffine;
  OR::Vector m_upRotAxis;
  DblVec m_upQ, m_upGrab;
  int m_upLinks = 0;                // nodes of the uploaded tree: the robot's, then one per grabbed link with geometry
  vector<const KinBody::Link*> m_upNodeLinks;   // the link each node stands for (NULL: an intermediate joint frame)
  OR::Transform m_upBase;
  vector<int> m_affectedObjs;       // object ids of the bound configuration's affected links with geometry
  unsigned long m_rebuilds = 0, m_boundRebuilds = ~0ul;   // scene uploads so far / the one m_affectedObjs was derived under
  DblVec m_boundGrab;               // the robot's grab set m_affectedObjs was derived under (GrabSignature)
  // prefetch
  vector<Prefetched> m_prefetch;
  std::unordered_multimap<size_t, int> m_prefetchIndex;
  short m_prefetchMask;
  long m_prefetchHits, m_engineCalls;
};

}  // namespace trajopt
