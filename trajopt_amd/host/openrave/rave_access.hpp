// rave_access.hpp (OpenRAVE implementation) -- the same declarations as tests/cpp/standin/rave_access.hpp, over
// OpenRAVE's public API (>= 0.9, doc/install.rst:12).  A build inside an OpenRAVE tree puts THIS directory on the include
// path instead of tests/cpp/standin (and <openrave/openrave.h> takes the place of rave_shim.hpp).
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: OpenRAVE is absent, so this file has never met a compiler; it
// is written against the headers of OpenRAVE 0.9 (kinbody.h, robot.h) and kept small so that it can be read against
// them.  What makes a mistake here fail loudly instead of bending gradients is the FK guard in
// HIPCollisionChecker::UploadRobot: the tree below, run through the engine's FK kernel, must reproduce every
// link->GetTransform() of the live robot to 1e-9, or the upload throws.
//
// OpenRAVE's forward kinematics for a joint between its hierarchy parent link P and child link C (kinbody.cpp,
// KinBody::SetDOFValues):   T_C = T_P * tLeft * Motion(axis, value) * tRight
// with tLeft = joint->GetInternalHierarchyLeftTransform(), tRight = ...RightTransform(), axis =
// GetInternalHierarchyAxis(0) (in the joint frame).  The engine's tree has T_node = T_parent * origin * Motion, no right
// transform, so a joint becomes TWO nodes: the joint frame (origin = tLeft, moving) and the child link's frame (a fixed
// node, origin = tRight, carrying the link).  Links without a parent joint (the base link, links welded by static
// joints) are fixed nodes at their pose in the parent's / the body's frame.
#pragma once
#include <map>
#include <stdexcept>
#include <vector>

#include <openrave/openrave.h>

#include "trajopt_hipcc.h"

namespace OR = OpenRAVE;

namespace trajopt {
namespace rave_access {

struct TreeNode {
  int parent;
  int type;
  OR::Transform origin;
  OR::Vector axis;
  int dof;
  double mult, offs;
  const OR::KinBody::Link* link;
};

inline void FlattenRobot(const OR::RobotBase& r, std::vector<TreeNode>& nodes) {
  nodes.clear();
  std::map<const OR::KinBody::Link*, int> nodeOf;
  const OR::Transform Tbody = r.GetTransform();   // = the pose of link 0
  auto fixedNode = [&](int parent, const OR::Transform& origin, const OR::KinBody::Link* link) {
    TreeNode n;
    n.parent = parent, n.type = TOHC_J_FIXED, n.origin = origin, n.axis = OR::Vector(0, 0, 1), n.dof = -1, n.mult = 1, n.offs = 0, n.link = link;
    nodes.push_back(n);
    if (link) nodeOf[link] = (int)nodes.size() - 1;
    return (int)nodes.size() - 1;
  };
  // the base link: OpenRAVE defines the body transform as the transform of the first link
  const OR::KinBody::LinkPtr base = r.GetLinks().at(0);
  fixedNode(-1, OR::Transform(), base.get());
  // joints in dependency order: a joint's hierarchy parent link has been placed before the joint is reached
  // (mimic joints and passive joints appear in GetDependencyOrderedJoints / GetPassiveJoints; both are walked)
  std::vector<OR::KinBody::JointPtr> joints = r.GetDependencyOrderedJoints();
  const std::vector<OR::KinBody::JointPtr>& passive = r.GetPassiveJoints();
  joints.insert(joints.end(), passive.begin(), passive.end());
  bool progress = true;
  std::vector<char> done(joints.size(), 0);
  while (progress) {
    progress = false;
    for (size_t k = 0; k < joints.size(); ++k) {
      if (done[k]) continue;
      const OR::KinBody::JointPtr& j = joints[k];
      const OR::KinBody::LinkPtr P = j->GetHierarchyParentLink(), C = j->GetHierarchyChildLink();
      if (!P || !C || !nodeOf.count(P.get()) || nodeOf.count(C.get())) continue;
      done[k] = 1, progress = true;
      if (j->GetDOF() > 1) throw std::runtime_error("HIPCollisionChecker: joint '" + j->GetName() + "' has more than one degree of freedom");
      TreeNode n;
      n.parent = nodeOf[P.get()], n.origin = j->GetInternalHierarchyLeftTransform(), n.axis = j->GetInternalHierarchyAxis(0), n.link = NULL;
      n.mult = 1, n.offs = 0, n.dof = -1, n.type = TOHC_J_FIXED;
      if (j->IsStatic()) {
        // welded: the joint frame never moves
      } else if (j->IsMimic(0)) {
        // value = f(values of other joints): only the linear form  value = mult * q[dof] + offs  is representable
        std::vector<OR::KinBody::Joint::MIMIC::DOFFormat> deps;
        j->GetMimicDOFIndices(deps, 0);
        if (deps.size() != 1) throw std::runtime_error("HIPCollisionChecker: mimic joint '" + j->GetName() + "' depends on " + std::to_string(deps.size()) + " joints (1 supported)");
        // finite-difference the equation at two points: exact for a linear equation, and checked to be one
        const int d = deps[0].dofindex;
        std::vector<OR::dReal> q;
        r.GetDOFValues(q);
        OR::KinBody::KinBodyStateSaver saver(const_cast<OR::RobotBase&>(r).shared_kinbody());
        std::vector<OR::dReal> q0 = q, q1 = q, q2 = q;
        q0[d] = 0, q1[d] = 0.5, q2[d] = 1;
        OR::RobotBase& rw = const_cast<OR::RobotBase&>(r);
        rw.SetDOFValues(q0, false); const double v0 = j->GetValue(0);
        rw.SetDOFValues(q1, false); const double v1 = j->GetValue(0);
        rw.SetDOFValues(q2, false); const double v2 = j->GetValue(0);
        if (fabs((v2 - v0) - 2 * (v1 - v0)) > 1e-9) throw std::runtime_error("HIPCollisionChecker: mimic joint '" + j->GetName() + "' is not linear in its master");
        n.dof = d, n.mult = v2 - v0, n.offs = v0;
        n.type = j->IsRevolute(0) ? TOHC_J_REVOLUTE : TOHC_J_PRISMATIC;
      } else {
        n.dof = j->GetDOFIndex();   // index into robot->GetDOFValues()
        n.type = j->IsRevolute(0) ? TOHC_J_REVOLUTE : (j->IsPrismatic(0) ? TOHC_J_PRISMATIC : -1);
        if (n.type < 0) throw std::runtime_error("HIPCollisionChecker: joint '" + j->GetName() + "' is neither revolute nor prismatic");
        if (n.dof < 0) n.type = TOHC_J_FIXED;   // a passive joint without a mimic equation stays at its current value: folded into the frame
        if (n.dof < 0 && j->GetValue(0) != 0) {
          OR::Transform m;
          if (j->IsRevolute(0)) m.rot = OR::geometry::quatFromAxisAngle(n.axis, j->GetValue(0));
          else m.trans = n.axis * j->GetValue(0);
          n.origin = n.origin * m;
        }
      }
      nodes.push_back(n);
      fixedNode((int)nodes.size() - 1, j->GetInternalHierarchyRightTransform(), C.get());
    }
  }
  // links no joint reaches (rigidly attached in the body frame)
  for (const OR::KinBody::LinkPtr& l : r.GetLinks())
    if (!nodeOf.count(l.get())) fixedNode(0, Tbody.inverse() * l->GetTransform(), l.get());
}

struct GrabbedLink {
  const OR::KinBody::Link* link;
  const OR::KinBody::Link* grabber;
  OR::Transform poseInGrabber;
};
inline void GrabbedLinks(const OR::RobotBase& r, std::vector<GrabbedLink>& out) {
  out.clear();
  std::vector<OR::KinBodyPtr> grabbed;
  r.GetGrabbed(grabbed);                                        // configuration_space.cpp:120-121
  for (const OR::KinBodyPtr& body : grabbed) {
    const OR::KinBody::LinkPtr grabber = r.IsGrabbing(body);    // :123
    if (!grabber) continue;
    const OR::Transform Tg = grabber->GetTransform().inverse();
    for (const OR::KinBody::LinkPtr& l : body->GetLinks()) {
      GrabbedLink gl;
      gl.link = l.get(), gl.grabber = grabber.get(), gl.poseInGrabber = Tg * l->GetTransform();   // grabbed bodies move with their link: world poses are current
      out.push_back(gl);
    }
  }
}

}  // namespace rave_access
}  // namespace trajopt
