// hip_collision_checker.cpp -- see hip_collision_checker.hpp.  Every device operation goes through the C-ABI of
// libtrajopt_hipcc.so; there is no CPU path here: without the library / a GPU, tohc_create fails and the constructor throws.
#include "hip_collision_checker.hpp"

#include "rave_access.hpp"   // the robot's kinematic tree and grabbed links (stand-in here; openrave/rave_access.hpp upstream)

#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <functional>

namespace trajopt {

using OR::RobotBase;
using OR::Transform;
using OR::Vector;

// ===================================================================================================================
// small helpers
// ===================================================================================================================
static void put7(const Transform& T, double* o) {   // OpenRAVE quaternions are w-first
  o[0] = T.rot.x, o[1] = T.rot.y, o[2] = T.rot.z, o[3] = T.rot.w, o[4] = T.trans.x, o[5] = T.trans.y, o[6] = T.trans.z;
}
static Transform get7(const double* o) { return Transform(Vector(o[0], o[1], o[2], o[3]), Vector(o[4], o[5], o[6])); }
static bool isIdentity(const Transform& T) {   // bullet_collision_checker.cpp:85-95
  const double e = 1e-6;
  return fabs(T.trans.x) < e && fabs(T.trans.y) < e && fabs(T.trans.z) < e && fabs(T.rot.x - 1) < e && fabs(T.rot.y) < e && fabs(T.rot.z) < e &&
         fabs(T.rot.w) < e;
}

// The one factory of the reference (collision_checker.hpp:87, body at bullet_collision_checker.cpp:1036-1039), with the
// switch SURVEY.md 5 asks for: TRAJOPT_COLLISION_CHECKER = hip (default) | bullet.  Upstream the second branch is the
// reference's own `new BulletCollisionChecker(env)`; that class needs OpenRAVE + the vendored Bullet and is not part of
// this build, so asking for it here is an error, never a silent substitute.
CollisionCheckerPtr CreateCollisionChecker(OR::EnvironmentBaseConstPtr env) {
  const char* which = getenv("TRAJOPT_COLLISION_CHECKER");
  if (which && std::string(which) == "bullet")
    throw std::runtime_error("TRAJOPT_COLLISION_CHECKER=bullet: the reference's BulletCollisionChecker is not part of this build");
  if (which && std::string(which) != "hip" && std::string(which) != "")
    throw std::runtime_error(std::string("TRAJOPT_COLLISION_CHECKER: unknown checker '") + which + "' (hip | bullet)");
  return CollisionCheckerPtr(new HIPCollisionChecker(env));
}

// ===================================================================================================================
// HIPCollisionChecker
// ===================================================================================================================
void HIPCollisionChecker::chk(int rc) const {
  if (rc != TOHC_OK) throw std::runtime_error(std::string("HIPCollisionChecker: ") + tohc_last_error(m_e));   // the reference throws too
}

HIPCollisionChecker::HIPCollisionChecker(OR::EnvironmentBaseConstPtr env)
    : CollisionChecker(env), m_e(NULL), m_contactDistance(.05), m_bound(NULL), m_upRobot(NULL), m_upAffine(0), m_prefetchMask(-1), m_prefetchHits(0),
      m_engineCalls(0) {
  int dev = 0;
  if (const char* s = getenv("TRAJOPT_HIP_DEVICE")) dev = atoi(s);
  if (tohc_create(&m_e, dev) != TOHC_OK) throw std::runtime_error("HIPCollisionChecker: no MI355X / HIP runtime (the engine has no CPU fallback)");
  SetContactDistance(.05);   // bullet_collision_checker.cpp:437
  UpdateFromRave();
}

HIPCollisionChecker::~HIPCollisionChecker() { tohc_destroy(m_e); }

void HIPCollisionChecker::SetContactDistance(float d) {   // :449-460
  m_contactDistance = d;
  chk(tohc_set_contact_distance(m_e, d));
  InvalidatePrefetch();
}

void HIPCollisionChecker::Rebuild(const vector<OR::KinBodyPtr>& bodies) {
  vector<int32_t> kind, voff, vcnt, childShape, objOff, objCnt, objCompound, objGroup;
  vector<float> param, vx, vy, vz;
  vector<double> childT, staticT;
  m_obj2link.clear(), m_link2obj.clear();
  for (const OR::KinBodyPtr& body : bodies) {
    for (const KinBody::LinkPtr& link : body->GetLinks()) {
      const auto& geoms = link->GetGeometries();
      if (geoms.empty()) continue;   // CollisionObjectFromLink returns no object, :218
      objOff.push_back((int32_t)childShape.size()), objCnt.push_back((int32_t)geoms.size());
      objCompound.push_back(!(geoms.size() == 1 && isIdentity(geoms[0]->GetTransform())));   // :224
      objGroup.push_back(body->IsRobot() ? RobotFilter : KinBodyFilter);                    // :538
      double t7[7];
      put7(link->GetTransform(), t7);
      staticT.insert(staticT.end(), t7, t7 + 7);
      m_link2obj[link.get()] = (int)m_obj2link.size();
      m_obj2link.push_back(link.get());
      for (const auto& g : geoms) {   // createShapePrimitive, :134-212
        float p[4] = {0, 0, 0, 0};
        voff.push_back((int32_t)vx.size());
        switch (g->GetType()) {
          case OR::GT_Box: kind.push_back(TOHC_BOX), p[0] = (float)g->GetBoxExtents().x, p[1] = (float)g->GetBoxExtents().y, p[2] = (float)g->GetBoxExtents().z; break;
          case OR::GT_Sphere: kind.push_back(TOHC_SPHERE), p[0] = (float)g->GetSphereRadius(); break;
          case OR::GT_Cylinder: kind.push_back(TOHC_CYLZ), p[0] = (float)g->GetCylinderRadius(), p[1] = (float)(g->GetCylinderHeight() / 2.); break;   // :153-158
          case OR::GT_TriMesh: {
            // :160-203: below 50 vertices the mesh's vertices are the hull's points, from 50 on btShapeHull's <= 42
            // support points (tohc_shape_hull restates btShapeHull::buildHull + HullLibrary)
            const OR::TriMesh& mesh = g->GetCollisionMesh();
            if (mesh.indices.size() < 3) throw std::runtime_error("HIPCollisionChecker: triangle mesh without triangles");   // assert, :162
            vector<float> mv;
            for (const Vector& v : mesh.vertices) mv.push_back((float)v.x), mv.push_back((float)v.y), mv.push_back((float)v.z);   // toBt, :166
            vector<int32_t> mi(mesh.indices.begin(), mesh.indices.begin() + mesh.indices.size() / 3 * 3);
            int n = tohc_shape_hull(mv.data(), (int)mesh.vertices.size(), mi.data(), (int)mi.size() / 3, NULL, 0);
            if (n < 1 || n > 64) throw std::runtime_error("HIPCollisionChecker: convex geometry needs 1..64 points (meshes of 50..64 vertices that btShapeHull cannot reduce excepted, the engine's hulls hold at most 64)");
            vector<float> hull((size_t)3 * n);
            tohc_shape_hull(mv.data(), (int)mesh.vertices.size(), mi.data(), (int)mi.size() / 3, hull.data(), n);
            kind.push_back(TOHC_HULL);
            for (int i = 0; i < n; ++i) vx.push_back(hull[3 * i]), vy.push_back(hull[3 * i + 1]), vz.push_back(hull[3 * i + 2]);
            break;
          }
          default: throw std::runtime_error("unrecognized collision shape type");   // :208
        }
        vcnt.push_back((int32_t)vx.size() - voff.back());
        param.insert(param.end(), p, p + 4);
        childShape.push_back((int32_t)kind.size() - 1);
        put7(g->GetTransform(), t7);
        childT.insert(childT.end(), t7, t7 + 7);
      }
    }
  }
  int n = (int)m_obj2link.size();
  ++m_rebuilds;   // object numbers may have changed: whatever was derived from them (m_affectedObjs) is stale
  m_upRobot = NULL;
  m_uploadedT = staticT;
  InvalidatePrefetch();
  m_allowed.assign((size_t)n * n, 1);   // SetLinkIndices, :606-614
  if (n == 0) {                         // an empty world: nothing to upload, every query answers "no contacts"
    return;
  }
  float dummy = 0;
  chk(tohc_upload_scene(m_e, (int)kind.size(), kind.data(), param.data(), voff.data(), vcnt.data(), (int)vx.size(), vx.empty() ? &dummy : vx.data(),
                        vy.empty() ? &dummy : vy.data(), vz.empty() ? &dummy : vz.data(), (int)childShape.size(), childShape.data(), childT.data(), n,
                        objOff.data(), objCnt.data(), objCompound.data(), objGroup.data(), staticT.data()));
  chk(tohc_set_allowed_matrix(m_e, n, m_allowed.data()));
  chk(tohc_set_contact_distance(m_e, (float)m_contactDistance));
}

void HIPCollisionChecker::UpdateAllowedCollisionMatrix() {   // :616-627
  int n = (int)m_obj2link.size();
  for (const LinkPair& p : m_excludedPairs) {
    auto a = m_link2obj.find(p.first), b = m_link2obj.find(p.second);
    if (a != m_link2obj.end() && b != m_link2obj.end()) m_allowed[(size_t)a->second * n + b->second] = m_allowed[(size_t)b->second * n + a->second] = 0;
  }
  if (n) chk(tohc_set_allowed_matrix(m_e, n, m_allowed.data()));
  InvalidatePrefetch();
}

void HIPCollisionChecker::UpdateFromRave() {   // UpdateBulletFromRave, :629-655
  vector<OR::KinBodyPtr> bodies, added;
  m_env->GetBodies(bodies);
  if (bodies != m_prevbodies) {
    for (auto& b : bodies)
      if (std::find(m_prevbodies.begin(), m_prevbodies.end(), b) == m_prevbodies.end()) added.push_back(b);
    if (m_bound && std::find(bodies.begin(), bodies.end(), std::static_pointer_cast<KinBody>(m_bound->robot)) == bodies.end()) m_bound = NULL;
    Rebuild(bodies);
    m_prevbodies = bodies;
    float old = (float)GetContactDistance();
    SetContactDistance(.1f);
    for (auto& b : added) IgnoreZeroStateSelfCollisions(b);
    SetContactDistance(old);
    UpdateAllowedCollisionMatrix();
    return;
  }
  // Queries on explicit link lists hand the engine every pose (CurrentPoses).  The linearizing queries run FK on the
  // device and take the poses of everything that is not a robot link from the scene upload: refresh it when one moved.
  std::set<const KinBody*> carried;   // grabbed bodies move with their robot: their poses come from FK, not from the upload
  for (const OR::KinBodyPtr& b : bodies)
    if (const RobotBase* rb = dynamic_cast<const RobotBase*>(b.get())) {
      vector<OR::KinBodyPtr> grabbed;
      rb->GetGrabbed(grabbed);
      for (const OR::KinBodyPtr& g : grabbed) carried.insert(g.get());
    }
  for (size_t o = 0; o < m_obj2link.size(); ++o) {
    if (m_obj2link[o]->GetParent()->IsRobot() || carried.count(m_obj2link[o]->GetParent())) continue;
    double t7[7];
    put7(m_obj2link[o]->GetTransform(), t7);
    if (memcmp(t7, &m_uploadedT[7 * o], sizeof t7)) {
      vector<uint8_t> allowed = m_allowed;
      Rebuild(bodies);
      m_allowed = allowed;
      chk(tohc_set_allowed_matrix(m_e, (int)m_obj2link.size(), m_allowed.data()));
      break;
    }
  }
}

void HIPCollisionChecker::ExcludeCollisionPair(const KinBody::Link& l0, const KinBody::Link& l1) {   // :342-346 (one direction, like the reference)
  m_excludedPairs.insert(LinkPair(&l0, &l1));
  auto a = m_link2obj.find(&l0), b = m_link2obj.find(&l1);
  int n = (int)m_obj2link.size();
  if (a != m_link2obj.end() && b != m_link2obj.end()) {
    m_allowed[(size_t)a->second * n + b->second] = 0;
    chk(tohc_set_allowed_matrix(m_e, n, m_allowed.data()));
    InvalidatePrefetch();
  }
}

void HIPCollisionChecker::IncludeCollisionPair(const KinBody::Link& l0, const KinBody::Link& l1) {   // :347-351
  m_excludedPairs.erase(LinkPair(&l0, &l1));
  auto a = m_link2obj.find(&l0), b = m_link2obj.find(&l1);
  int n = (int)m_obj2link.size();
  if (a != m_link2obj.end() && b != m_link2obj.end()) {
    m_allowed[(size_t)a->second * n + b->second] = 1;
    chk(tohc_set_allowed_matrix(m_e, n, m_allowed.data()));
    InvalidatePrefetch();
  }
}

// ---- robot ---------------------------------------------------------------------------------------------------------
bool HIPCollisionChecker::RobotUploadIsCurrent(RobotAndDOF& rad) const {
  RobotBase* r = rad.robot.get();
  if (m_upRobot != r || m_upJointInds != rad.joint_inds || m_upAffine != rad.affinedofs) return false;
  if ((rad.affinedofs & OR::DOF_RotationAxis) &&
      (m_upRotAxis.x != rad.rotationaxis.x || m_upRotAxis.y != rad.rotationaxis.y || m_upRotAxis.z != rad.rotationaxis.z))
    return false;
  if (GrabSignature(*r) != m_upGrab) return false;
  const DblVec& q = r->GetDOFValues();
  for (size_t j = 0; j < q.size(); ++j)   // the joints that are not optimised keep the values of the upload
    if (q[j] != m_upQ[j] && std::find(rad.joint_inds.begin(), rad.joint_inds.end(), (int)j) == rad.joint_inds.end()) return false;
  if (rad.affinedofs == 0) {
    double a[7], b[7];
    put7(r->GetTransform(), a), put7(m_upBase, b);
    if (memcmp(a, b, sizeof a)) return false;
  }
  return true;
}

// what of the robot's grabbed bodies the upload depends on: (link, grabbing link, relative pose) of each grabbed link
vector<double> HIPCollisionChecker::GrabSignature(const RobotBase& r) const {
  vector<double> sig;
  vector<rave_access::GrabbedLink> gl;
  rave_access::GrabbedLinks(r, gl);
  for (const rave_access::GrabbedLink& g : gl) {
    double t7[7];
    put7(g.poseInGrabber, t7);
    sig.push_back((double)(uintptr_t)g.link), sig.push_back((double)(uintptr_t)g.grabber);
    sig.insert(sig.end(), t7, t7 + 7);
  }
  return sig;
}

// ---- affine DOFs <-> body transform (RaveGetTransformFromAffineDOFValues / RaveGetAffineDOFValuesFromTransform, as
// RobotAndDOF uses them, configuration_space.cpp:12-35) ----------------------------------------------------------------
static Transform affineTransform(const double* v, int affinedofs, const Vector& axis) {
  Transform T;
  if (affinedofs & OR::DOF_X) T.trans.x = *v++;
  if (affinedofs & OR::DOF_Y) T.trans.y = *v++;
  if (affinedofs & OR::DOF_Z) T.trans.z = *v++;
  if (affinedofs & OR::DOF_RotationAxis) {
    const double a = *v++, s = sin(a / 2);
    T.rot = Vector(cos(a / 2), axis.x * s, axis.y * s, axis.z * s);
  }
  return T;
}
static double poseDistance(const Transform& a, const Transform& b) {   // max |difference| of translation and (sign-free) quaternion
  double d = std::max(fabs(a.trans.x - b.trans.x), std::max(fabs(a.trans.y - b.trans.y), fabs(a.trans.z - b.trans.z)));
  const double sgn = (a.rot.x * b.rot.x + a.rot.y * b.rot.y + a.rot.z * b.rot.z + a.rot.w * b.rot.w) < 0 ? -1 : 1;
  d = std::max(d, std::max(std::max(fabs(a.rot.x - sgn * b.rot.x), fabs(a.rot.y - sgn * b.rot.y)), std::max(fabs(a.rot.z - sgn * b.rot.z), fabs(a.rot.w - sgn * b.rot.w))));
  return d;
}
// Does the configuration's DOF vector describe the robot's state?  Without affine DOFs always (the body transform is part
// of the upload); with them only while the body transform is one the affine DOFs can express -- IgnoreZeroStateSelfCollisions
// lifts the body to z = 10 by SetTransform (collision_checker.cpp:47), which DOF_X | DOF_Y | DOF_RotationAxis cannot.
bool HIPCollisionChecker::StateIsRepresentable(RobotAndDOF& rad) const {
  if (rad.affinedofs == 0) return true;
  const DblVec q = rad.GetDOFValues();
  return poseDistance(affineTransform(q.data() + rad.joint_inds.size(), rad.affinedofs, rad.rotationaxis), rad.robot->GetTransform()) <= 1e-12;
}

void HIPCollisionChecker::UploadRobot(RobotAndDOF& rad) {
  if (RobotUploadIsCurrent(rad)) return;
  RobotBase* r = rad.robot.get();
  vector<rave_access::TreeNode> nodes;
  rave_access::FlattenRobot(*r, nodes);
  const int nr = (int)nodes.size();
  std::map<const KinBody::Link*, int> nodeOf;
  for (int i = 0; i < nr; ++i)
    if (nodes[i].link) nodeOf[nodes[i].link] = i;
  // Grabbed bodies (configuration_space.cpp:120-131): every link with geometry of a grabbed body becomes a FIXED node
  // hanging off the grabbing link, marked TOHC_DOF_GRABBED -- a query link whatever moves it, with the grabber's Jacobian
  vector<rave_access::GrabbedLink> grabbed, extra;
  rave_access::GrabbedLinks(*r, grabbed);
  for (const rave_access::GrabbedLink& g : grabbed)
    if (!g.link->GetGeometries().empty() && m_link2obj.count(g.link) && nodeOf.count(g.grabber)) extra.push_back(g);
  int n = nr + (int)extra.size();
  vector<int32_t> parent(n), jtype(n), dof(n), linkObj(n);
  vector<double> origin(7 * n), axis(3 * n), mult(n), offs(n);
  m_upNodeLinks.assign(n, (const KinBody::Link*)NULL);
  for (int k = 0; k < (int)extra.size(); ++k) {
    const int i = nr + k;
    parent[i] = nodeOf.at(extra[k].grabber), jtype[i] = TOHC_J_FIXED, dof[i] = TOHC_DOF_GRABBED, mult[i] = 1, offs[i] = 0;
    put7(extra[k].poseInGrabber, &origin[7 * i]);
    axis[3 * i] = 0, axis[3 * i + 1] = 0, axis[3 * i + 2] = 1;
    linkObj[i] = m_link2obj.at(extra[k].link);
    m_upNodeLinks[i] = extra[k].link;
  }
  for (int i = 0; i < nr; ++i) {
    const rave_access::TreeNode& J = nodes[i];
    parent[i] = J.parent, jtype[i] = J.type, dof[i] = J.dof, mult[i] = J.mult, offs[i] = J.offs;
    put7(J.origin, &origin[7 * i]);
    double l = sqrt(J.axis.x * J.axis.x + J.axis.y * J.axis.y + J.axis.z * J.axis.z);
    axis[3 * i] = J.axis.x / l, axis[3 * i + 1] = J.axis.y / l, axis[3 * i + 2] = J.axis.z / l;
    auto it = J.link ? m_link2obj.find(J.link) : m_link2obj.end();
    linkObj[i] = it == m_link2obj.end() ? -1 : it->second;
    m_upNodeLinks[i] = J.link;
  }
  DblVec q = r->GetDOFValues();
  double base[7], rot[3] = {rad.rotationaxis.x, rad.rotationaxis.y, rad.rotationaxis.z};
  put7(r->GetTransform(), base);
  vector<int32_t> ji(rad.joint_inds.begin(), rad.joint_inds.end());
  double qdummy = 0;
  chk(tohc_upload_robot(m_e, n, parent.data(), jtype.data(), origin.data(), axis.data(), dof.data(), mult.data(), offs.data(), linkObj.data(),
                        (int)q.size(), q.empty() ? &qdummy : q.data(), base, (int)ji.size(), ji.data(), rad.affinedofs, rot));
  m_upRobot = r, m_upJointInds = rad.joint_inds, m_upAffine = rad.affinedofs, m_upRotAxis = rad.rotationaxis, m_upQ = q, m_upBase = r->GetTransform();
  m_upGrab = GrabSignature(*r), m_upLinks = n;
  CheckDeviceFK(rad);
}

// The FK guard.  The linearizing queries run FK on the DEVICE, on the tree rave_access handed over; the explicit-pose
// queries take link->GetTransform(), the host's FK (OpenRAVE's).  Contacts and gradients are only meaningful if the two
// agree, and nothing in the reference pins the conventions (OpenRAVE is an external dependency: SURVEY.md 8c), so every
// upload is checked at the robot's current state: each node that stands for a link, run through tohc_forward_kinematics,
// must land on that link's GetTransform() -- relative to the base link always, absolutely whenever the DOF vector can
// express the body transform.  A wrong joint convention then fails here, at bind time, instead of bending gradients.
void HIPCollisionChecker::CheckDeviceFK(RobotAndDOF& rad) {
  const double tol = 1e-9;
  const bool absolute = StateIsRepresentable(rad);
  DblVec q = rad.GetDOFValues();
  if (!absolute) std::fill(q.begin() + rad.joint_inds.size(), q.end(), 0.0);   // (any base will do for the relative check)
  vector<double> nodeT(7 * (size_t)m_upLinks);
  chk(tohc_forward_kinematics(m_e, q.data(), 1, nodeT.data()));
  ++m_engineCalls;
  int ref = -1;
  for (int i = 0; i < m_upLinks && ref < 0; ++i)
    if (m_upNodeLinks[i]) ref = i;
  if (ref < 0) return;
  const Transform devRefInv = get7(&nodeT[7 * ref]).inverse(), hostRefInv = m_upNodeLinks[ref]->GetTransform().inverse();
  for (int i = 0; i < m_upLinks; ++i) {
    if (!m_upNodeLinks[i]) continue;
    const Transform dev = get7(&nodeT[7 * i]), host = m_upNodeLinks[i]->GetTransform();
    const double dr = poseDistance(devRefInv * dev, hostRefInv * host), da = absolute ? poseDistance(dev, host) : 0;
    if (!(dr <= tol) || !(da <= tol)) {
      m_upRobot = NULL;   // this upload must not be used
      char msg[512];
      snprintf(msg, sizeof msg, "HIPCollisionChecker: the engine's forward kinematics disagrees with link '%s' (GetTransform) by %.3g relative to the base link"
               "%s: the kinematic tree handed to tohc_upload_robot does not describe this robot", m_upNodeLinks[i]->GetName().c_str(), dr,
               absolute ? (", " + std::to_string(da) + " absolutely").c_str() : "");
      throw std::runtime_error(msg);
    }
  }
}

void HIPCollisionChecker::BindConfiguration(Configuration& c) {
  RobotAndDOF* rad = dynamic_cast<RobotAndDOF*>(&c);
  if (!rad) throw std::runtime_error("HIPCollisionChecker: only RobotAndDOF configurations are supported");
  UpdateFromRave();
  // The kept binding stands while the configuration is the same AND what m_affectedObjs was derived from is: the object
  // numbering (a Rebuild renumbers) and the robot's grab set (GetAffectedLinks lists grabbed links, configuration_space.cpp:
  // 120-131) -- a Grab / Release between two evaluators on the same robot and joints must refresh the list, or LinksVsAll /
  // CastVsAll would take the explicit-pose route, whose contacts carry no linearization.
  const vector<double> grabNow = GrabSignature(*rad->robot);
  if (m_boundKeep.get() && m_bound == m_boundKeep.get() && m_bound->robot == rad->robot && m_bound->joint_inds == rad->joint_inds &&
      m_bound->affinedofs == rad->affinedofs && m_bound->rotationaxis.x == rad->rotationaxis.x && m_bound->rotationaxis.y == rad->rotationaxis.y &&
      m_bound->rotationaxis.z == rad->rotationaxis.z && m_boundRebuilds == m_rebuilds && m_boundGrab == grabNow)
    return;
  // keep our own copy: evaluators come and go, the checker lives with the environment
  m_boundKeep.reset(new RobotAndDOF(rad->robot, rad->joint_inds, rad->affinedofs, rad->rotationaxis));
  m_bound = m_boundKeep.get();
  m_boundRebuilds = m_rebuilds, m_boundGrab = grabNow;
  vector<KinBody::LinkPtr> links;
  vector<int> inds;
  m_bound->GetAffectedLinks(links, true, inds);
  m_affectedObjs.clear();
  for (auto& l : links) m_affectedObjs.push_back(m_link2obj.at(l.get()));
  std::sort(m_affectedObjs.begin(), m_affectedObjs.end());
  InvalidatePrefetch();
}

bool HIPCollisionChecker::IsAffectedSet(const vector<KinBody::LinkPtr>& links) const {
  vector<int> ids;
  for (auto& l : links) {
    auto it = m_link2obj.find(l.get());
    if (it != m_link2obj.end()) ids.push_back(it->second);
  }
  std::sort(ids.begin(), ids.end());
  return ids == m_affectedObjs;
}

vector<double> HIPCollisionChecker::CurrentPoses() {   // UpdateBulletFromRave's pose loop, :648-653
  vector<double> T(7 * m_obj2link.size());
  for (size_t o = 0; o < m_obj2link.size(); ++o) put7(m_obj2link[o]->GetTransform(), &T[7 * o]);
  return T;
}

Collision HIPCollisionChecker::ToCollision(const tohc_record& r, const double* grads, int K) const {
  Collision c(m_obj2link.at(r.idA), m_obj2link.at(r.idB), Vector(r.ptA[0], r.ptA[1], r.ptA[2]), Vector(r.ptB[0], r.ptB[1], r.ptB[2]),
              Vector(r.n[0], r.n[1], r.n[2]), r.dist);
  c.ptB1 = Vector(r.ptB1[0], r.ptB1[1], r.ptB1[2]);
  c.time = r.time, c.cctype = (CastCollisionType)r.cctype, c.childA = r.childA, c.childB = r.childB;
  c.const0 = r.const0, c.const1 = r.const1;
  if (grads) {
    c.grad0.assign(grads, grads + K);
    if (r.cctype != CCType_None) c.grad1.assign(grads + K, grads + 2 * K);
  }
  return c;
}

// ---- queries -------------------------------------------------------------------------------------------------------
const HIPCollisionChecker::Prefetched* HIPCollisionChecker::FindPrefetched(const DblVec& q0, const DblVec* q1, short filterMask) const {
  if (m_prefetch.empty() || (!q1 && filterMask != m_prefetchMask)) return NULL;
  if (q1) {   // CastVsAll is handed its two DOF vectors: exact look-up
    size_t key = hashRange(q0, 1);
    auto range = m_prefetchIndex.equal_range(key);
    for (auto it = range.first; it != range.second; ++it) {
      const Prefetched& p = m_prefetch[it->second];
      if (p.q0 == q0 && p.q1 == *q1) return &p;
    }
    return NULL;
  }
  // LinksVsAll is handed nothing: its DOF vector is read back from the robot (RobotAndDOF::GetDOFValues), and an affine
  // rotation angle does not survive angle -> quaternion -> angle bit for bit.  So: the prefetched timestep within 1e-12 of
  // the robot's state (far below any step the optimiser takes; the contacts handed out are those AT the prefetched vector).
  for (const Prefetched& p : m_prefetch) {
    if (!p.q1.empty() || p.q0.size() != q0.size()) continue;
    double d = 0;
    for (size_t i = 0; i < q0.size() && d <= 1e-12; ++i) d = std::max(d, fabs(p.q0[i] - q0[i]));
    if (d <= 1e-12) return &p;
  }
  return NULL;
}

void HIPCollisionChecker::LinksVsAll(const vector<KinBody::LinkPtr>& links, vector<Collision>& collisions, short filterMask) {   // :500-510
  UpdateFromRave();
  if (m_obj2link.empty()) return;
  // The linearizing route runs FK on the device from the bound configuration's DOF vector, so it only applies while
  // that vector describes the robot's state (StateIsRepresentable; else the explicit-pose route below).
  if (m_bound && IsAffectedSet(links) && StateIsRepresentable(*m_bound)) {
    // the query the collision terms make (collision_terms.cpp:118-122): contacts AND their linearization
    DblVec q = m_bound->GetDOFValues();
    if (const Prefetched* p = FindPrefetched(q, NULL, filterMask)) {
      collisions.insert(collisions.end(), p->collisions.begin(), p->collisions.end());
      ++m_prefetchHits;
      return;
    }
    UploadRobot(*m_bound);
    const tohc_record* r;
    const double* g;
    int n, K = m_bound->GetDOF();
    chk(tohc_links_vs_all_batch(m_e, q.data(), 1, 1, filterMask, 1, &r, &g, &n));
    ++m_engineCalls;
    for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], g + (size_t)i * 2 * K, K));   // appended, never cleared (:404)
    return;
  }
  vector<int32_t> query;
  for (auto& l : links) {
    auto it = m_link2obj.find(l.get());
    if (it != m_link2obj.end()) query.push_back(it->second);   // links without geometry: LinkVsAll_NoUpdate returns at once (:520)
  }
  if (query.empty()) return;
  vector<double> T = CurrentPoses();
  const tohc_record* r;
  int n;
  chk(tohc_links_vs_all_poses(m_e, T.data(), 1, query.data(), (int)query.size(), filterMask, &r, &n));
  ++m_engineCalls;
  for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], NULL, 0));
}

void HIPCollisionChecker::LinkVsAll(const KinBody::Link& link, vector<Collision>& collisions, short filterMask) {   // :513-516
  UpdateFromRave();
  auto it = m_link2obj.find(&link);
  if (it == m_link2obj.end()) return;
  vector<double> T = CurrentPoses();
  int32_t q = it->second;
  const tohc_record* r;
  int n;
  chk(tohc_links_vs_all_poses(m_e, T.data(), 1, &q, 1, filterMask, &r, &n));
  ++m_engineCalls;
  for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], NULL, 0));
}

void HIPCollisionChecker::AllVsAll(vector<Collision>& collisions) {   // :463-498 ("seems to be broken" upstream)
  UpdateFromRave();
  if (m_obj2link.empty()) return;
  vector<double> T = CurrentPoses();
  const tohc_record* r;
  int n;
  chk(tohc_all_vs_all_poses(m_e, T.data(), &r, &n));   // one manifold point per convex pair: weight 1. / numContacts = 1 (:487)
  ++m_engineCalls;
  for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], NULL, 0));
}

void HIPCollisionChecker::CastVsAll(Configuration& rad, const vector<KinBody::LinkPtr>& links, const DblVec& startjoints, const DblVec& endjoints,
                                    vector<Collision>& collisions) {   // :1000-1023
  BindConfiguration(rad);
  if (m_obj2link.empty()) return;
  if (IsAffectedSet(links)) {
    if (const Prefetched* p = FindPrefetched(startjoints, &endjoints, -1)) {
      collisions.insert(collisions.end(), p->collisions.begin(), p->collisions.end());
      ++m_prefetchHits;
      rad.SetDOFValues(startjoints);   // the reference leaves the robot at the start (:1013)
      return;
    }
    rad.SetDOFValues(startjoints);
    UploadRobot(*m_bound);
    int K = m_bound->GetDOF();
    DblVec dofs(startjoints);
    dofs.insert(dofs.end(), endjoints.begin(), endjoints.end());
    const tohc_record* r;
    const double* g;
    int n;
    chk(tohc_cast_vs_all_batch(m_e, dofs.data(), 1, 2, 1, 1, &r, &g, &n));
    ++m_engineCalls;
    for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], g + (size_t)i * 2 * K, K));
    return;
  }
  // any other link list: sweep exactly those links between their poses at the two configurations
  vector<int32_t> query;
  vector<const KinBody::Link*> qlinks;
  for (auto& l : links) {
    auto it = m_link2obj.find(l.get());
    if (it != m_link2obj.end()) query.push_back(it->second), qlinks.push_back(l.get());
  }
  if (query.empty()) return;
  rad.SetDOFValues(endjoints);
  vector<double> T1(7 * query.size());
  for (size_t i = 0; i < qlinks.size(); ++i) put7(qlinks[i]->GetTransform(), &T1[7 * i]);
  rad.SetDOFValues(startjoints);
  vector<double> T0 = CurrentPoses();
  const tohc_record* r;
  int n;
  chk(tohc_cast_vs_all_poses(m_e, T0.data(), T1.data(), 1, query.data(), (int)query.size(), &r, &n));
  ++m_engineCalls;
  for (int i = 0; i < n; ++i) collisions.push_back(ToCollision(r[i], NULL, 0));
}

// bullet_collision_checker.cpp:352-360: UpdateBulletFromRave + rayTest with a ClosestRayResultCallback -> hasHit()
vector<float> HIPCollisionChecker::RayCastCollisions(const vector<double>& points1, const vector<double>& points2) {
  if (points1.size() != points2.size() || points1.size() % 3 != 0) throw std::runtime_error("RayCastCollisions: point arrays must hold the same number of xyz triples");
  vector<float> closest(points1.size() / 3, 1.0f);
  UpdateFromRave();   // :356 -- bodies added, removed or moved since the last call
  if (closest.empty() || m_obj2link.empty()) return closest;   // an empty world: nothing to hit
  vector<double> T = CurrentPoses();
  chk(tohc_ray_cast_poses(m_e, T.data(), points1.data(), points2.data(), (int)closest.size(), closest.data()));
  ++m_engineCalls;
  return closest;
}
bool HIPCollisionChecker::RayCastCollision(const OR::Vector& point1, const OR::Vector& point2) {
  const vector<double> a = {point1.x, point1.y, point1.z}, b = {point2.x, point2.y, point2.z};
  return RayCastCollisions(a, b)[0] < 1.0f;
}

void HIPCollisionChecker::ContinuousCheckTrajectory(const TrajArray& traj, Configuration& rad, vector<Collision>& collisions) {   // :708-753
  BindConfiguration(rad);
  if (m_obj2link.empty() || traj.rows() < 2) return;
  if (traj.cols() != m_bound->GetDOF()) throw std::runtime_error("ContinuousCheckTrajectory: trajectory width != DOF");
  UploadRobot(*m_bound);
  const tohc_record* r;
  int n;
  chk(tohc_continuous_check_trajectory(m_e, traj.data(), 1, traj.rows(), &r, &n));
  ++m_engineCalls;
  vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  // the reference's order: link, geometry, segment (ContinuousCheckShape's recursion, :683-705)
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (r[a].idA != r[b].idA) return r[a].idA < r[b].idA;
    if (r[a].childA != r[b].childA) return r[a].childA < r[b].childA;
    return r[a].t < r[b].t;
  });
  for (int i : order) {
    Vector hit(r[i].ptA[0], r[i].ptA[1], r[i].ptA[2]);
    Collision c(m_obj2link.at(r[i].idA), m_obj2link.at(r[i].idB), hit, hit, Vector(r[i].n[0], r[i].n[1], r[i].n[2]), 0, 1, r[i].t + r[i].time);
    c.childA = r[i].childA, c.childB = r[i].childB;
    c.const1 = (double)((float)r[i].t + r[i].time);   // the time of impact the reference's constructor drops (collision_checker.hpp:28)
    collisions.push_back(c);
  }
}

void HIPCollisionChecker::PrefetchTrajectory(Configuration& rad, const TrajArray& traj, bool discrete, bool cast, short filterMask) {
  BindConfiguration(rad);
  InvalidatePrefetch();
  int T = traj.rows(), K = m_bound->GetDOF();
  if (m_obj2link.empty() || T < 1 || !(discrete || cast)) return;
  if (traj.cols() != K) throw std::runtime_error("PrefetchTrajectory: trajectory width != DOF");
  cast = cast && T >= 2;
  UploadRobot(*m_bound);
  chk(tohc_stage_dofs(m_e, traj.data(), 1, T));
  if (discrete && cast) chk(tohc_enqueue_linearization(m_e, filterMask, 1, 1));   // both passes, one broadphase launch
  else if (discrete) chk(tohc_enqueue_links_vs_all(m_e, filterMask, 1, 0));
  else if (cast) chk(tohc_enqueue_cast_vs_all(m_e, 1, 1, 0));
  const tohc_record* r;
  const double* g;
  int n;
  chk(tohc_fetch(m_e, &r, &g, &n));
  ++m_engineCalls;
  m_prefetchMask = filterMask;
  // one entry per timestep / per pair, contacts or not: "no contacts" is an answer too
  int nd = discrete ? T : 0, nc = cast ? T - 1 : 0;
  m_prefetch.resize(nd + nc);
  for (int t = 0; t < nd; ++t) m_prefetch[t].q0 = traj.row(t);
  for (int t = 0; t < nc; ++t) m_prefetch[nd + t].q0 = traj.row(t), m_prefetch[nd + t].q1 = traj.row(t + 1);
  for (int i = 0; i < n; ++i) {
    bool isCast = r[i].cctype != CCType_None;   // CastCollisionCollector always sets Time0 / Time1 / Between (:909-967)
    m_prefetch[(isCast ? nd : 0) + r[i].t].collisions.push_back(ToCollision(r[i], g + (size_t)i * 2 * K, K));
  }
  for (int i = nd; i < nd + nc; ++i) m_prefetchIndex.insert(std::make_pair(hashRange(m_prefetch[i].q0, 1), i));   // (cast pairs; discrete entries are scanned)
}

}  // namespace trajopt
