// trajopt_host.cpp -- see trajopt_host.hpp.  Every device operation goes through the C-ABI of libtrajopt_hipcc.so;
// there is no CPU path here: without the library / a GPU, tohc_create fails and the constructor throws.
#include "trajopt_host.hpp"

#include <algorithm>
#include <cassert>
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
static size_t hashRange(const DblVec& x, size_t seed = 0) {   // boost::hash_range (collision_terms.cpp:82-84)
  std::hash<double> h;
  for (double v : x) seed ^= h(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
  return seed;
}
static int affineCount(int a) { return ((a & 1) != 0) + ((a & 2) != 0) + ((a & 4) != 0) + ((a & 8) != 0); }

std::ostream& operator<<(std::ostream& o, const Collision& c) {   // collision_checker.cpp:68-77
  auto v = [&](const Vector& a) { o << "[" << a.x << ", " << a.y << ", " << a.z << "]"; };
  o << (c.linkA ? c.linkA->GetName() : "NULL") << "--" << (c.linkB ? c.linkB->GetName() : "NULL") << " distance: " << c.distance << " normal: ";
  v(c.normalB2A); o << " ptA: "; v(c.ptA); o << " ptB: "; v(c.ptB);
  o << " time: " << c.time << " weight: " << c.weight;
  return o;
}

// ===================================================================================================================
// CollisionChecker (collision_checker.cpp)
// ===================================================================================================================
std::shared_ptr<CollisionChecker> CollisionChecker::GetOrCreate(OR::EnvironmentBase& env) {   // :16-27
  OR::UserDataPtr ud = env.m_userdata["trajopt_cc"];
  if (!ud) {
    ud = CreateCollisionChecker(env.shared_from_this());
    env.m_userdata["trajopt_cc"] = ud;
  }
  return std::dynamic_pointer_cast<CollisionChecker>(ud);
}

void CollisionChecker::IgnoreZeroStateSelfCollisions(OR::KinBodyPtr body) {   // :41-56
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

void CollisionChecker::IgnoreZeroStateSelfCollisions() {   // :58-66
  vector<OR::KinBodyPtr> bodies;
  GetEnv()->GetBodies(bodies);
  for (auto& b : bodies) IgnoreZeroStateSelfCollisions(b);
}

CollisionCheckerPtr CreateCollisionChecker(OR::EnvironmentBaseConstPtr env) {   // bullet_collision_checker.cpp:1036-1039
  return CollisionCheckerPtr(new HIPCollisionChecker(env));
}

// ===================================================================================================================
// RobotAndDOF (configuration_space.cpp)
// ===================================================================================================================
int RobotAndDOF::GetDOF() const { return (int)joint_inds.size() + affineCount(affinedofs); }   // :80-82

void RobotAndDOF::SetDOFValues(const DblVec& dofs) {   // :12-23
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
  robot->CacheAffineDOFValues(affinedofs, rotationaxis, DblVec(dofs.begin() + joint_inds.size(), dofs.end()));
}

DblVec RobotAndDOF::GetDOFValues() {   // :25-35
  DblVec out, aff;
  for (int j : joint_inds) out.push_back(robot->GetDOFValues().at(j));
  if (affinedofs != 0 && robot->GetCachedAffineDOFValues(affinedofs, rotationaxis, aff)) {
    out.insert(out.end(), aff.begin(), aff.end());
  } else if (affinedofs != 0) {   // RaveGetAffineDOFValuesFromTransform
    const Transform& T = robot->GetTransform();
    if (affinedofs & OR::DOF_X) out.push_back(T.trans.x);
    if (affinedofs & OR::DOF_Y) out.push_back(T.trans.y);
    if (affinedofs & OR::DOF_Z) out.push_back(T.trans.z);
    if (affinedofs & OR::DOF_RotationAxis)
      out.push_back(2 * atan2(T.rot.y * rotationaxis.x + T.rot.z * rotationaxis.y + T.rot.w * rotationaxis.z, T.rot.x));
  }
  return out;
}

bool RobotAndDOF::DoesAffect(const KinBody::Link& link) {   // :97-101, rave_utils.cpp:69-74
  if (affinedofs > 0) return true;
  if (link.GetParent() != robot.get()) return false;
  const vector<RobotBase::Joint>& J = robot->GetJoints();
  for (int i = link.GetIndex(); i >= 0; i = J[i].parent)
    if (J[i].type != OR::JointFixed && std::find(joint_inds.begin(), joint_inds.end(), J[i].dof) != joint_inds.end()) return true;
  return false;
}

void RobotAndDOF::GetAffectedLinks(vector<KinBody::LinkPtr>& links, bool only_with_geom, vector<int>& link_inds) {   // :110-118
  links.clear(), link_inds.clear();
  for (const KinBody::LinkPtr& link : robot->GetLinks())
    if (DoesAffect(*link) && !(only_with_geom && link->GetGeometries().empty())) links.push_back(link), link_inds.push_back(link->GetIndex());
  // grabbed bodies (:120-131): the shim has no Grab(); SURVEY.md 8(f3)
}

// ===================================================================================================================
// HIPCollisionChecker
// ===================================================================================================================
void HIPCollisionChecker::chk(int rc) const {
  if (rc != TOHC_OK) throw std::runtime_error(std::string("HIPCollisionChecker: ") + tohc_last_error(m_e));   // the reference throws too
}

HIPCollisionChecker::HIPCollisionChecker(OR::EnvironmentBaseConstPtr env)
    : CollisionChecker(env), m_e(NULL), m_contactDistance(.05), m_bound(NULL), m_fkRobot(NULL), m_upRobot(NULL), m_upAffine(0), m_fkStamp(~0ul),
      m_prefetchMask(-1), m_prefetchHits(0), m_engineCalls(0) {
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
      objCompound.push_back(!(geoms.size() == 1 && isIdentity(geoms[0].GetTransform())));   // :224
      objGroup.push_back(body->IsRobot() ? RobotFilter : KinBodyFilter);                    // :538
      double t7[7];
      put7(link->GetTransform(), t7);
      staticT.insert(staticT.end(), t7, t7 + 7);
      m_link2obj[link.get()] = (int)m_obj2link.size();
      m_obj2link.push_back(link.get());
      for (const auto& g : geoms) {   // createShapePrimitive, :134-212
        float p[4] = {0, 0, 0, 0};
        voff.push_back((int32_t)vx.size());
        switch (g.GetType()) {
          case OR::GT_Box: kind.push_back(TOHC_BOX), p[0] = (float)g.boxExtents.x, p[1] = (float)g.boxExtents.y, p[2] = (float)g.boxExtents.z; break;
          case OR::GT_Sphere: kind.push_back(TOHC_SPHERE), p[0] = (float)g.radius; break;
          case OR::GT_Cylinder: kind.push_back(TOHC_CYLZ), p[0] = (float)g.radius, p[1] = (float)(g.height / 2.); break;   // :153-158
          case OR::GT_TriMesh:
            // the engine takes the FINAL hull points: meshes of >= 50 vertices are reduced by btShapeHull in the
            // reference (:174-203); that reduction is host-side set-up the caller runs once
            if (g.meshVertices.empty() || g.meshVertices.size() > 64) throw std::runtime_error("HIPCollisionChecker: convex geometry needs 1..64 points");
            kind.push_back(TOHC_HULL);
            for (const Vector& v : g.meshVertices) vx.push_back((float)v.x), vy.push_back((float)v.y), vz.push_back((float)v.z);
            break;
          default: throw std::runtime_error("unrecognized collision shape type");   // :208
        }
        vcnt.push_back((int32_t)vx.size() - voff.back());
        param.insert(param.end(), p, p + 4);
        childShape.push_back((int32_t)kind.size() - 1);
        put7(g.GetTransform(), t7);
        childT.insert(childT.end(), t7, t7 + 7);
      }
    }
  }
  int n = (int)m_obj2link.size();
  m_upRobot = NULL, m_fkRobot = NULL, m_fkStamp = ~0ul;
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
  for (size_t o = 0; o < m_obj2link.size(); ++o) {
    if (m_obj2link[o]->GetParent()->IsRobot()) continue;
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

void HIPCollisionChecker::UploadRobot(RobotAndDOF& rad) {
  if (RobotUploadIsCurrent(rad)) return;
  RobotBase* r = rad.robot.get();
  const vector<RobotBase::Joint>& J = r->GetJoints();
  int n = (int)J.size();
  vector<int32_t> parent(n), jtype(n), dof(n), linkObj(n);
  vector<double> origin(7 * n), axis(3 * n), mult(n), offs(n);
  for (int i = 0; i < n; ++i) {
    parent[i] = J[i].parent, jtype[i] = J[i].type, dof[i] = J[i].dof, mult[i] = J[i].mult, offs[i] = J[i].offs;
    put7(J[i].origin, &origin[7 * i]);
    double l = sqrt(J[i].axis.x * J[i].axis.x + J[i].axis.y * J[i].axis.y + J[i].axis.z * J[i].axis.z);
    axis[3 * i] = J[i].axis.x / l, axis[3 * i + 1] = J[i].axis.y / l, axis[3 * i + 2] = J[i].axis.z / l;
    auto it = m_link2obj.find(r->GetLinks()[i].get());
    linkObj[i] = it == m_link2obj.end() ? -1 : it->second;
  }
  DblVec q = r->GetDOFValues();
  double base[7], rot[3] = {rad.rotationaxis.x, rad.rotationaxis.y, rad.rotationaxis.z};
  put7(r->GetTransform(), base);
  vector<int32_t> ji(rad.joint_inds.begin(), rad.joint_inds.end());
  double qdummy = 0;
  chk(tohc_upload_robot(m_e, n, parent.data(), jtype.data(), origin.data(), axis.data(), dof.data(), mult.data(), offs.data(), linkObj.data(),
                        (int)q.size(), q.empty() ? &qdummy : q.data(), base, (int)ji.size(), ji.data(), rad.affinedofs, rot));
  m_upRobot = r, m_upJointInds = rad.joint_inds, m_upAffine = rad.affinedofs, m_upRotAxis = rad.rotationaxis, m_upQ = q, m_upBase = r->GetTransform();
}

void HIPCollisionChecker::BindConfiguration(Configuration& c) {
  RobotAndDOF* rad = dynamic_cast<RobotAndDOF*>(&c);
  if (!rad) throw std::runtime_error("HIPCollisionChecker: only RobotAndDOF configurations are supported");
  UpdateFromRave();
  if (m_boundKeep.get() && m_bound == m_boundKeep.get() && m_bound->robot == rad->robot && m_bound->joint_inds == rad->joint_inds &&
      m_bound->affinedofs == rad->affinedofs && m_bound->rotationaxis.x == rad->rotationaxis.x && m_bound->rotationaxis.y == rad->rotationaxis.y &&
      m_bound->rotationaxis.z == rad->rotationaxis.z)
    return;
  // keep our own copy: evaluators come and go, the checker lives with the environment
  m_boundKeep.reset(new RobotAndDOF(rad->robot, rad->joint_inds, rad->affinedofs, rad->rotationaxis));
  m_bound = m_boundKeep.get();
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

void HIPCollisionChecker::UpdateRobotLinkTransforms() {
  for (const OR::KinBodyPtr& body : m_prevbodies) {
    RobotBase* r = dynamic_cast<RobotBase*>(body.get());
    if (!r) continue;
    const vector<RobotBase::Joint>& J = r->GetJoints();
    if (r->GetDOF() == 0) {   // a rigid robot (boxbot): compose on the host, nothing for the FK kernel to do
      for (size_t i = 0; i < J.size(); ++i)
        r->GetLinks()[i]->m_T = (J[i].parent < 0 ? r->GetTransform() : r->GetLinks()[J[i].parent]->m_T) * J[i].origin;
      continue;
    }
    if (m_fkRobot == r && m_fkStamp == r->GetStamp()) continue;
    // all joints active, base = the body transform: the engine's FK kernel run on the robot's current state
    IntVec all(r->GetDOF());
    for (int j = 0; j < r->GetDOF(); ++j) all[j] = j;
    RobotAndDOF full(std::dynamic_pointer_cast<RobotBase>(body), all, 0);
    UploadRobot(full);
    vector<double> linkT(7 * J.size());
    chk(tohc_forward_kinematics(m_e, r->GetDOFValues().data(), 1, linkT.data()));
    ++m_engineCalls;
    for (size_t i = 0; i < J.size(); ++i) r->GetLinks()[i]->m_T = get7(&linkT[7 * i]);
    m_fkRobot = r, m_fkStamp = r->GetStamp();
  }
}

vector<double> HIPCollisionChecker::CurrentPoses() {
  UpdateRobotLinkTransforms();
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
  size_t key = hashRange(q0, q1 ? 1 : 0);
  auto range = m_prefetchIndex.equal_range(key);
  for (auto it = range.first; it != range.second; ++it) {
    const Prefetched& p = m_prefetch[it->second];
    if (p.q0 == q0 && (q1 ? p.q1 == *q1 : p.q1.empty())) return &p;
  }
  return NULL;
}

void HIPCollisionChecker::LinksVsAll(const vector<KinBody::LinkPtr>& links, vector<Collision>& collisions, short filterMask) {   // :500-510
  UpdateFromRave();
  if (m_obj2link.empty()) return;
  // The linearizing route runs FK on the device from the bound configuration's DOF vector, so it only applies while
  // that vector describes the robot's state: with affine DOFs, while the body transform is the one last set through a
  // RobotAndDOF (IgnoreZeroStateSelfCollisions lifts the body to z = 10 by SetTransform: explicit-pose route then).
  DblVec aff;
  const bool representable = m_bound && (m_bound->affinedofs == 0 || m_bound->robot->GetCachedAffineDOFValues(m_bound->affinedofs, m_bound->rotationaxis, aff));
  if (representable && IsAffectedSet(links)) {
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
  vector<int32_t> query(m_obj2link.size());
  for (size_t i = 0; i < query.size(); ++i) query[i] = (int32_t)i;
  const tohc_record* r;
  int n;
  chk(tohc_links_vs_all_poses(m_e, T.data(), 1, query.data(), (int)query.size(), -1, &r, &n));
  ++m_engineCalls;
  std::set<std::pair<std::pair<int, int>, std::pair<int, int> > > seen;   // each unordered geometry pair once; weight 1/numContacts = 1
  for (int i = 0; i < n; ++i) {
    std::pair<int, int> a(r[i].idA, r[i].childA), b(r[i].idB, r[i].childB);
    if (b < a) std::swap(a, b);
    if (seen.insert(std::make_pair(a, b)).second) collisions.push_back(ToCollision(r[i], NULL, 0));
  }
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
  UpdateRobotLinkTransforms();
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
  if (closest.empty()) return closest;
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
  if (discrete) chk(tohc_enqueue_links_vs_all(m_e, filterMask, 1, 0));
  if (cast) chk(tohc_enqueue_cast_vs_all(m_e, 1, 1, discrete ? 1 : 0));
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
  for (int i = 0; i < nd + nc; ++i) m_prefetchIndex.insert(std::make_pair(hashRange(m_prefetch[i].q0, m_prefetch[i].q1.empty() ? 0 : 1), i));
}

// ===================================================================================================================
// collision_terms.cpp
// ===================================================================================================================
void CollisionsToDistances(const vector<Collision>& collisions, const Link2Int& link2ind, DblVec& dists) {   // :22-35
  dists.clear();
  dists.reserve(collisions.size());
  for (const Collision& col : collisions)
    if (link2ind.find(col.linkA) != link2ind.end() || link2ind.find(col.linkB) != link2ind.end()) dists.push_back(col.distance);
}

// :37-62.  The reference computes, per robot-side link, dist_grad = +-n^T J(link, pt) through PositionJacobian and adds
// varDot(dist_grad, vars) - dist_grad . dofvals.  The kernels did exactly that sum over both sides in k_finish:
// grad = sum of the sides' rows, const = distance - grad . dofvals.  So the expression is assembled, not recomputed.
void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration&, const Link2Int& link2ind, const VarVector& vars,
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

void CollisionsToDistanceExpressions(const vector<Collision>& collisions, Configuration& rad, const Link2Int& link2ind, const VarVector& vars0,
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

void CollisionEvaluator::GetCollisionsCached(const DblVec& x, vector<Collision>& collisions) {   // :86-98
  size_t key = hashRange(getDblVec(x, GetVars()));
  vector<Collision>* it = m_cache.get(key);
  if (it != NULL) {
    collisions = *it;
  } else {
    CalcCollisions(x, collisions);
    m_cache.put(key, collisions);
  }
}

static void bind(CollisionCheckerPtr cc, Configuration& rad) {
  // the one line the reference's evaluators do not have: tell the checker which configuration LinksVsAll linearizes
  if (HIPCollisionChecker* h = dynamic_cast<HIPCollisionChecker*>(cc.get())) h->BindConfiguration(rad);
}

SingleTimestepCollisionEvaluator::SingleTimestepCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars)   // :100-115
    : m_env(rad->GetEnv()), m_cc(CollisionChecker::GetOrCreate(*m_env)), m_rad(rad), m_vars(vars), m_filterMask(-1) {
  vector<int> inds;
  rad->GetAffectedLinks(m_links, true, inds);
  for (size_t i = 0; i < m_links.size(); ++i) m_link2ind[m_links[i].get()] = inds[i];
}

void SingleTimestepCollisionEvaluator::CalcCollisions(const DblVec& x, vector<Collision>& collisions) {   // :118-122
  DblVec dofvals = getDblVec(x, m_vars);
  bind(m_cc, *m_rad);
  m_rad->SetDOFValues(dofvals);
  m_cc->LinksVsAll(m_links, collisions, m_filterMask);
}

void SingleTimestepCollisionEvaluator::CalcDists(const DblVec& x, DblVec& dists) {   // :124-128
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  CollisionsToDistances(collisions, m_link2ind, dists);
}

void SingleTimestepCollisionEvaluator::CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs) {   // :131-136
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  DblVec dofvals = getDblVec(x, m_vars);
  CollisionsToDistanceExpressions(collisions, *m_rad, m_link2ind, m_vars, dofvals, exprs, false);
}

CastCollisionEvaluator::CastCollisionEvaluator(ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :140-154
    : m_env(rad->GetEnv()), m_cc(CollisionChecker::GetOrCreate(*m_env)), m_rad(rad), m_vars0(vars0), m_vars1(vars1), m_filterMask(-1) {
  vector<int> inds;
  rad->GetAffectedLinks(m_links, true, inds);
  for (size_t i = 0; i < m_links.size(); ++i) m_link2ind[m_links[i].get()] = inds[i];
}

void CastCollisionEvaluator::CalcCollisions(const DblVec& x, vector<Collision>& collisions) {   // :156-161
  DblVec dofvals0 = getDblVec(x, m_vars0);
  DblVec dofvals1 = getDblVec(x, m_vars1);
  m_rad->SetDOFValues(dofvals0);
  m_cc->CastVsAll(*m_rad, m_links, dofvals0, dofvals1, collisions);
}

void CastCollisionEvaluator::CalcDistExpressions(const DblVec& x, vector<AffExpr>& exprs) {   // :162-168
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  DblVec dofvals0 = getDblVec(x, m_vars0);
  DblVec dofvals1 = getDblVec(x, m_vars1);
  CollisionsToDistanceExpressions(collisions, *m_rad, m_link2ind, m_vars0, m_vars1, dofvals0, dofvals1, exprs);
}

void CastCollisionEvaluator::CalcDists(const DblVec& x, DblVec& dists) {   // :169-173
  vector<Collision> collisions;
  GetCollisionsCached(x, collisions);
  CollisionsToDistances(collisions, m_link2ind, dists);
}

CollisionCost::CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars)   // :196-199
    : m_calc(new SingleTimestepCollisionEvaluator(rad, vars)), m_dist_pen(dist_pen), m_coeff(coeff) {}
CollisionCost::CollisionCost(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :201-204
    : m_calc(new CastCollisionEvaluator(rad, vars0, vars1)), m_dist_pen(dist_pen), m_coeff(coeff) {}

ConvexObjectivePtr CollisionCost::convex(const DblVec& x) {   // :205-214
  ConvexObjectivePtr out(new ConvexObjective);
  vector<AffExpr> exprs;
  m_calc->CalcDistExpressions(x, exprs);
  for (size_t i = 0; i < exprs.size(); ++i) {
    AffExpr viol = exprSub(AffExpr(m_dist_pen), exprs[i]);
    out->addHinge(viol, m_coeff);
  }
  return out;
}

double CollisionCost::value(const DblVec& x) {   // :215-223
  DblVec dists;
  m_calc->CalcDists(x, dists);
  double out = 0;
  for (size_t i = 0; i < dists.size(); ++i) out += pospart(m_dist_pen - dists[i]) * m_coeff;
  return out;
}

CollisionConstraint::CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars)   // :233-237
    : m_calc(new SingleTimestepCollisionEvaluator(rad, vars)), m_dist_pen(dist_pen), m_coeff(coeff) {}
CollisionConstraint::CollisionConstraint(double dist_pen, double coeff, ConfigurationPtr rad, const VarVector& vars0, const VarVector& vars1)   // :239-243
    : m_calc(new CastCollisionEvaluator(rad, vars0, vars1)), m_dist_pen(dist_pen), m_coeff(coeff) {}

ConvexConstraintsPtr CollisionConstraint::convex(const DblVec& x) {   // :244-253
  ConvexConstraintsPtr out(new ConvexConstraints);
  vector<AffExpr> exprs;
  m_calc->CalcDistExpressions(x, exprs);
  for (size_t i = 0; i < exprs.size(); ++i) {
    AffExpr viol = exprSub(AffExpr(m_dist_pen), exprs[i]);
    out->addIneqCnt(exprMult(viol, m_coeff));
  }
  return out;
}

DblVec CollisionConstraint::value(const DblVec& x) {   // :254-262
  DblVec dists;
  m_calc->CalcDists(x, dists);
  DblVec out(dists.size());
  for (size_t i = 0; i < dists.size(); ++i) out[i] = pospart(m_dist_pen - dists[i]) * m_coeff;
  return out;
}

}  // namespace trajopt
