// rave_shim.hpp -- the handful of OpenRAVE types the collision path mentions, as plain C++.
//
// OpenRAVE (an un-vendored dependency of the reference, CMakeLists.txt:22-24) is not in this image, so the host
// layer (trajopt_amd/host/hip_collision_checker.hpp) cannot be compiled against it here.  This header declares stand-ins with the same names
// and accessors that src/trajopt/{collision_checker.hpp, bullet_collision_checker.cpp, configuration_space.cpp,
// collision_terms.cpp} use (KinBody, KinBody::Link, Link::Geometry, RobotBase, EnvironmentBase, Vector, Transform),
// so that host code and its tests read like the reference's.  Inside a real OpenRAVE process this header is replaced
// by <openrave/openrave.h>.  Like OpenRAVE, the stand-in keeps every link's world pose current (RobotBase runs its own
// fp64 host FK over a flattened joint tree); what the checker needs beyond OpenRAVE's public accessors -- the kinematic
// tree in the layout tohc_upload_robot takes, the grabbed bodies' poses in their grabbers' frames -- it reads through
// rave_access.hpp (here: over this stand-in; upstream: trajopt_amd/host/openrave/rave_access.hpp over OpenRAVE's API).
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace OR {

struct Vector {
  double x, y, z, w;
  Vector() : x(0), y(0), z(0), w(0) {}
  Vector(double x_, double y_, double z_, double w_ = 0) : x(x_), y(y_), z(z_), w(w_) {}
};
inline Vector operator+(const Vector& a, const Vector& b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline Vector operator-(const Vector& a, const Vector& b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline Vector operator*(double s, const Vector& a) { return Vector(s * a.x, s * a.y, s * a.z, s * a.w); }
inline Vector operator*(const Vector& a, double s) { return s * a; }

// rot is a quaternion stored OpenRAVE style: (x, y, z, w) members hold (qw, qx, qy, qz)
struct Transform {
  Vector rot, trans;
  Transform() : rot(1, 0, 0, 0) {}
  Transform(const Vector& r, const Vector& t) : rot(r), trans(t) {}
  Vector rotate(const Vector& v) const {   // q v q*
    double w = rot.x, a = rot.y, b = rot.z, c = rot.w;
    double tx = 2 * (b * v.z - c * v.y), ty = 2 * (c * v.x - a * v.z), tz = 2 * (a * v.y - b * v.x);
    return Vector(v.x + w * tx + (b * tz - c * ty), v.y + w * ty + (c * tx - a * tz), v.z + w * tz + (a * ty - b * tx));
  }
  Vector operator*(const Vector& v) const { return rotate(v) + trans; }
  Transform inverse() const {
    Transform r;
    r.rot = Vector(rot.x, -rot.y, -rot.z, -rot.w);
    const Vector t = r.rotate(trans);
    r.trans = Vector(-t.x, -t.y, -t.z);
    return r;
  }
  bool isIdentity() const { return rot.x == 1 && rot.y == 0 && rot.z == 0 && rot.w == 0 && trans.x == 0 && trans.y == 0 && trans.z == 0; }
  Transform operator*(const Transform& o) const {
    if (o.isIdentity()) return *this;   // keep poses bit-exact through the common single-link case
    if (isIdentity()) return o;
    double w1 = rot.x, x1 = rot.y, y1 = rot.z, z1 = rot.w, w2 = o.rot.x, x2 = o.rot.y, y2 = o.rot.z, z2 = o.rot.w;
    Vector q(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
             w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2);
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Transform(Vector(q.x / n, q.y / n, q.z / n, q.w / n), rotate(o.trans) + trans);
  }
};

enum GeometryType { GT_None = 0, GT_Box = 1, GT_Sphere = 2, GT_Cylinder = 3, GT_TriMesh = 4 };
enum DOFAffine { DOF_NoTransform = 0, DOF_X = 1, DOF_Y = 2, DOF_Z = 4, DOF_RotationAxis = 8 };
enum JointType { JointFixed = 0, JointRevolute = 1, JointPrismatic = 2 };

class KinBody;
class EnvironmentBase;

struct TriMesh {   // OpenRAVE::TriMesh: GetCollisionMesh() of a GT_TriMesh geometry
  std::vector<Vector> vertices;
  std::vector<int> indices;   // three per triangle
};

class KinBody {
 public:
  class Link {
   public:
    struct Geometry {
      GeometryType type = GT_None;
      Vector boxExtents;               // GT_Box: half extents
      double radius = 0, height = 0;   // GT_Sphere / GT_Cylinder (full height)
      TriMesh mesh;                    // GT_TriMesh
      Transform t;                     // geometry frame in the link frame
      // the accessors createShapePrimitive uses (bullet_collision_checker.cpp:134-212)
      GeometryType GetType() const { return type; }
      const Transform& GetTransform() const { return t; }
      const Vector& GetBoxExtents() const { return boxExtents; }
      double GetSphereRadius() const { return radius; }
      double GetCylinderRadius() const { return radius; }
      double GetCylinderHeight() const { return height; }
      const TriMesh& GetCollisionMesh() const { return mesh; }
    };
    typedef std::shared_ptr<Geometry> GeometryPtr;
    const std::string& GetName() const { return m_name; }
    // a link's world pose is always current, as in OpenRAVE (which runs FK inside SetDOFValues / SetTransform): robots
    // refresh their links on demand (RobotBase::RefreshLinks)
    const Transform& GetTransform() const;
    const std::vector<GeometryPtr>& GetGeometries() const { return m_geoms; }
    GeometryPtr GetGeometry(int i) const { return m_geoms.at(i); }
    KinBody* GetParent() const { return m_parent; }
    int GetIndex() const { return m_index; }
    void AddGeometry(const Geometry& g) { m_geoms.push_back(GeometryPtr(new Geometry(g))); }   // (shim only: OpenRAVE builds links from XML / GeometryInfo)

    std::string m_name;
    std::vector<GeometryPtr> m_geoms;
    mutable Transform m_T;   // world pose
    Transform m_rel;         // pose in the body frame (rigid bodies)
    KinBody* m_parent = nullptr;
    int m_index = 0;
  };
  typedef std::shared_ptr<Link> LinkPtr;

  explicit KinBody(const std::string& name) : m_name(name) {}
  virtual ~KinBody() {}
  virtual bool IsRobot() const { return false; }
  virtual int GetDOF() const { return 0; }
  const std::string& GetName() const { return m_name; }
  void SetName(const std::string& n) { m_name = n; }
  const std::vector<LinkPtr>& GetLinks() const { return m_links; }
  const Transform& GetTransform() const { return m_T; }
  virtual void SetTransform(const Transform& T) {
    m_T = T;
    for (auto& l : m_links) l->m_T = T * l->m_rel;
  }
  virtual void RefreshLinks() const {}   // rigid bodies: SetTransform has placed the links
  LinkPtr AddLink(const std::string& name, const Transform& rel = Transform()) {
    LinkPtr l(new Link);
    l->m_name = name, l->m_rel = rel, l->m_T = m_T * rel, l->m_parent = this, l->m_index = (int)m_links.size();
    m_links.push_back(l);
    return l;
  }

 protected:
  std::string m_name;
  std::vector<LinkPtr> m_links;
  Transform m_T;
};
typedef std::shared_ptr<KinBody> KinBodyPtr;
inline const Transform& KinBody::Link::GetTransform() const {
  if (m_parent) m_parent->RefreshLinks();
  return m_T;
}

// A robot as a flattened joint tree, parents first: link i hangs off link parent by the joint stored with it;
// T_world_link = T_world_parent * origin * Motion(type, axis, mult * q[dof] + offs)   (DESIGN.md "FK model").
class RobotBase : public KinBody {
 public:
  struct Joint {
    int parent = -1;
    JointType type = JointFixed;
    Transform origin;
    Vector axis = Vector(0, 0, 1);
    int dof = -1;
    double mult = 1, offs = 0;
  };
  explicit RobotBase(const std::string& name) : KinBody(name) {}
  bool IsRobot() const override { return true; }
  int GetDOF() const override { return (int)m_dofvalues.size(); }
  LinkPtr AddLink(const std::string& name, const Joint& j) {
    if (j.parent >= (int)m_links.size()) throw std::runtime_error("RobotBase::AddLink: parents must precede children");
    LinkPtr l = KinBody::AddLink(name);
    m_joints.push_back(j);
    if (j.dof >= (int)m_dofvalues.size()) m_dofvalues.resize(j.dof + 1, 0.0);
    return l;
  }
  void SetTransform(const Transform& T) override { m_T = T, Touch(); }
  // Host FK, the stand-in for OpenRAVE's own (the checker never runs FK for the host: it reads link->GetTransform() like
  // the reference does, bullet_collision_checker.cpp:252,652).  fp64, parents first:
  //   T_link = T_parent * origin * Motion(type, axis, mult * q[dof] + offs);  grabbed bodies ride on their grabbing link.
  void RefreshLinks() const override {
    if (m_fkStamp == m_stamp) return;
    m_fkStamp = m_stamp;
    for (size_t i = 0; i < m_joints.size(); ++i) {
      const Joint& j = m_joints[i];
      Transform motion;
      if (j.type != JointFixed && j.dof >= 0) {
        const double a = j.mult * m_dofvalues[j.dof] + j.offs;
        const double n = std::sqrt(j.axis.x * j.axis.x + j.axis.y * j.axis.y + j.axis.z * j.axis.z);
        if (j.type == JointRevolute) { const double s = std::sin(a / 2) / n; motion.rot = Vector(std::cos(a / 2), j.axis.x * s, j.axis.y * s, j.axis.z * s); }
        else motion.trans = Vector(j.axis.x * a / n, j.axis.y * a / n, j.axis.z * a / n);
      }
      m_links[i]->m_T = ((j.parent < 0 ? m_T : m_links[j.parent]->m_T) * j.origin) * motion;
    }
    for (const Grabbed& g : m_grabbed) g.body->SetTransform(m_links[g.link]->m_T * g.rel);
  }
  void SetDOFValues(const std::vector<double>& v, const std::vector<int>& inds) {
    for (size_t i = 0; i < inds.size(); ++i) m_dofvalues.at(inds[i]) = v.at(i);
    Touch();
  }
  void SetDOFValues(const std::vector<double>& v) { m_dofvalues = v, Touch(); }
  const std::vector<double>& GetDOFValues() const { return m_dofvalues; }
  const std::vector<Joint>& GetJoints() const { return m_joints; }
  // Grab / Release / GetGrabbed / IsGrabbing as OpenRAVE has them (configuration_space.cpp:120-131 uses the last two).
  // The grabbed body keeps its pose relative to the grabbing link: `rel` = link^-1 * body at the time of the grab (the
  // link's pose is current then, like any link pose); GetGrabbedInfo is the shim's window on it (OpenRAVE:
  // RobotBase::GetGrabbedInfo, _trelative) -- read only by tests/cpp/standin/rave_access.hpp.
  struct Grabbed { std::shared_ptr<KinBody> body; int link; Transform rel; };
  void Grab(std::shared_ptr<KinBody> body, LinkPtr link) {
    Release(body);
    Grabbed g;
    g.body = body, g.link = link->GetIndex(), g.rel = link->GetTransform().inverse() * body->GetTransform();
    m_grabbed.push_back(g);
    Touch();
  }
  // the same with the relative pose given (OpenRAVE: ResetGrabbed with a GrabbedInfo::_trelative)
  void GrabAt(std::shared_ptr<KinBody> body, LinkPtr link, const Transform& rel) {
    Release(body);
    Grabbed g;
    g.body = body, g.link = link->GetIndex(), g.rel = rel;
    m_grabbed.push_back(g);
    Touch();
  }
  void Release(std::shared_ptr<KinBody> body) {
    for (size_t i = 0; i < m_grabbed.size(); ++i)
      if (m_grabbed[i].body == body) { m_grabbed.erase(m_grabbed.begin() + i); Touch(); return; }
  }
  void GetGrabbed(std::vector<std::shared_ptr<KinBody> >& out) const {
    out.clear();
    for (const Grabbed& g : m_grabbed) out.push_back(g.body);
  }
  LinkPtr IsGrabbing(const std::shared_ptr<KinBody>& body) const {
    for (const Grabbed& g : m_grabbed)
      if (g.body == body) return m_links[g.link];
    return LinkPtr();
  }
  const std::vector<Grabbed>& GetGrabbedInfo() const { return m_grabbed; }

 private:
  void Touch() { ++m_stamp, RefreshLinks(); }   // eager, like OpenRAVE: grabbed bodies move the moment the robot does
  std::vector<Joint> m_joints;
  std::vector<Grabbed> m_grabbed;
  std::vector<double> m_dofvalues;
  unsigned long m_stamp = 0;
  mutable unsigned long m_fkStamp = ~0ul;
};
typedef std::shared_ptr<RobotBase> RobotBasePtr;

class UserData {
 public:
  virtual ~UserData() {}
};
typedef std::shared_ptr<UserData> UserDataPtr;

class EnvironmentBase : public std::enable_shared_from_this<EnvironmentBase> {
 public:
  void Add(KinBodyPtr b) { m_bodies.push_back(b); }
  void Remove(KinBodyPtr b) {
    for (size_t i = 0; i < m_bodies.size(); ++i)
      if (m_bodies[i] == b) { m_bodies.erase(m_bodies.begin() + i); return; }
  }
  void GetBodies(std::vector<KinBodyPtr>& out) const { out = m_bodies; }
  KinBodyPtr GetKinBody(const std::string& name) const {
    for (auto& b : m_bodies) if (b->GetName() == name) return b;
    return KinBodyPtr();
  }
  RobotBasePtr GetRobot(const std::string& name) const { return std::dynamic_pointer_cast<RobotBase>(GetKinBody(name)); }
  std::map<std::string, UserDataPtr> m_userdata;   // openrave_userdata_utils.hpp

 private:
  std::vector<KinBodyPtr> m_bodies;
};
typedef std::shared_ptr<EnvironmentBase> EnvironmentBasePtr;
typedef std::shared_ptr<const EnvironmentBase> EnvironmentBaseConstPtr;

}  // namespace OR
