// ref_driver.cpp -- OpenRAVE-free driver around the REFERENCE's own vendored, patched, fp32 Bullet.
//
// TEST INFRASTRUCTURE ONLY (see oracle_api.h).  Built by oracle/Makefile in the container that has
// /root/reference; it #includes Bullet headers from /root/reference/ext/bullet/src and links the object
// compiled from /root/reference/src/trajopt/bullet_unity.cpp.  Output goes to oracle/_ref/ only.
//
// libtrajopt itself cannot be compiled here (needs OpenRAVE/Boost/Eigen), so the ~300 lines of glue in
// src/trajopt/bullet_collision_checker.cpp are restated over the flat ora_scene:
//   world set-up              bullet_collision_checker.cpp:427-439
//   SetContactDistance        :449-460
//   shape / object creation   :134-255 (box, hull, sphere, cylinder-Z; bare shape vs compound)
//   LinksVsAll                :500-524  + CollisionCollector :391-414
//   CastVsAll/CheckShapeCast  :974-1023 + CastHullShape :816-882 + CastCollisionCollector :894-972
//   GetAverageSupport         :98-131
#include <btBulletCollisionCommon.h>
#include <BulletCollision/CollisionDispatch/btConvexConvexAlgorithm.h>
#include <BulletCollision/CollisionShapes/btShapeHull.h>
#include <BulletCollision/CollisionShapes/btConvexPolyhedron.h>
#include <LinearMath/btAabbUtil2.h>
#include <vector>
#include <stdexcept>
#include <cstring>
#include <cmath>
#include "oracle_api.h"

extern btScalar gContactBreakingThreshold;
extern int gNumGjkChecks;
extern int gNumDeepPenetrationChecks;

namespace {

struct Obj : public btCollisionObject {
  int id;     // index into ora_scene objects == row of the allowed matrix
  Obj(int i) : id(i) {}
};

btTransform toBt7(const double* t) {
  // OR::Transform -> btTransform: every component rounded to float, quaternion reordered to (x,y,z,w)
  return btTransform(btQuaternion((btScalar)t[1], (btScalar)t[2], (btScalar)t[3], (btScalar)t[0]),
                     btVector3((btScalar)t[4], (btScalar)t[5], (btScalar)t[6]));
}

struct World;

struct Collector : public btCollisionWorld::ContactResultCallback {
  std::vector<ora_contact>& out;
  const Obj* self;
  World* w;
  Collector(std::vector<ora_contact>& o, const Obj* s, World* ww) : out(o), self(s), w(ww) {}
  virtual btScalar addSingleResult(btManifoldPoint& cp, const btCollisionObjectWrapper* o0, int, int index0,
                                   const btCollisionObjectWrapper* o1, int, int index1);
  virtual bool needsCollision(btBroadphaseProxy* proxy0) const;
};

struct SweptHull : public btConvexShape {  // restates CastHullShape
  btConvexShape* inner;
  btTransform t01;
  SweptHull(btConvexShape* s, const btTransform& t) : inner(s), t01(t) { m_shapeType = CUSTOM_CONVEX_SHAPE_TYPE; }
  btVector3 localGetSupportingVertex(const btVector3& v) const {
    btVector3 s0 = inner->localGetSupportingVertex(v);
    btVector3 s1 = t01 * inner->localGetSupportingVertex(v * t01.getBasis());
    return (v.dot(s0) > v.dot(s1)) ? s0 : s1;
  }
  btVector3 localGetSupportingVertexWithoutMargin(const btVector3& v) const { return localGetSupportingVertex(v); }
  void batchedUnitVectorGetSupportingVertexWithoutMargin(const btVector3*, btVector3*, int) const {
    throw std::runtime_error("not implemented");
  }
  void getAabb(const btTransform& tw0, btVector3& mn, btVector3& mx) const {
    inner->getAabb(tw0, mn, mx);
    btVector3 mn1, mx1;
    inner->getAabb(tw0 * t01, mn1, mx1);
    mn.setMin(mn1);
    mx.setMax(mx1);
  }
  void getAabbSlow(const btTransform&, btVector3&, btVector3&) const { throw std::runtime_error("no"); }
  void setLocalScaling(const btVector3&) {}
  const btVector3& getLocalScaling() const { static btVector3 one(1, 1, 1); return one; }
  void setMargin(btScalar) {}
  btScalar getMargin() const { return 0; }
  int getNumPreferredPenetrationDirections() const { return 0; }
  void getPreferredPenetrationDirection(int, btVector3&) const { throw std::runtime_error("no"); }
  void calculateLocalInertia(btScalar, btVector3&) const { throw std::runtime_error("no"); }
  const char* getName() const { return "SweptHull"; }
};

struct World {
  btDefaultCollisionConfiguration* cfg;
  btCollisionDispatcher* disp;
  btDbvtBroadphase* bp;
  btCollisionWorld* world;
  std::vector<Obj*> objs;
  std::vector<btCollisionShape*> owned;
  std::vector<unsigned char> allowed;
  int n;
  double contactDistance;

  void setContactDistance(float d) {
    contactDistance = d;
    SHAPE_EXPANSION = btVector3(1, 1, 1) * d;
    gContactBreakingThreshold = 2.001 * d;
    btCollisionObjectArray& a = world->getCollisionObjectArray();
    for (int i = 0; i < a.size(); ++i) a[i]->setContactProcessingThreshold(d);
    disp->setDispatcherFlags(disp->getDispatcherFlags() & ~btCollisionDispatcher::CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD);
  }
  bool canCollide(int a, int b) const { return allowed[(size_t)a * n + b] != 0; }
};

btScalar Collector::addSingleResult(btManifoldPoint& cp, const btCollisionObjectWrapper* o0, int, int index0,
                                    const btCollisionObjectWrapper* o1, int, int index1) {
  if (cp.m_distance1 > w->contactDistance) return 0;
  ora_contact c;
  memset(&c, 0, sizeof c);
  const Obj* A = static_cast<const Obj*>(o0->getCollisionObject());
  const Obj* B = static_cast<const Obj*>(o1->getCollisionObject());
  c.objA = A->id;
  c.objB = B->id;
  c.childA = A->getCollisionShape()->isCompound() ? index0 : 0;
  c.childB = B->getCollisionShape()->isCompound() ? index1 : 0;
  for (int k = 0; k < 3; ++k) {
    c.ptA[k] = cp.m_positionWorldOnA[k];
    c.ptB[k] = cp.m_positionWorldOnB[k];
    c.n[k] = cp.m_normalWorldOnB[k];
  }
  c.dist = cp.m_distance1;
  c.time = 0;
  c.cctype = ORA_CC_NONE;
  out.push_back(c);
  return 1;
}

bool Collector::needsCollision(btBroadphaseProxy* p) const {
  return (p->m_collisionFilterGroup & m_collisionFilterMask) && (m_collisionFilterGroup & p->m_collisionFilterMask) &&
         w->canCollide(self->id, static_cast<Obj*>(p->m_clientObject)->id);
}

void averageSupport(const btConvexShape* shape, const btVector3& localNormal, float& outsupport, btVector3& outpt) {
  btVector3 ptSum(0, 0, 0);
  float ptCount = 0;
  float maxSupport = -1000;
  const float EPSILON = 1e-3;
  const btPolyhedralConvexShape* ps = dynamic_cast<const btPolyhedralConvexShape*>(shape);
  if (ps) {
    int nPts = ps->getNumVertices();
    for (int i = 0; i < nPts; ++i) {
      btVector3 pt;
      ps->getVertex(i, pt);
      float sup = pt.dot(localNormal);
      if (sup > maxSupport + EPSILON) {
        ptCount = 1;
        ptSum = pt;
        maxSupport = sup;
      } else if (sup < maxSupport - EPSILON) {
      } else {
        ptCount += 1;
        ptSum += pt;
      }
    }
    outsupport = maxSupport;
    outpt = ptSum / ptCount;
  } else {
    outpt = shape->localGetSupportingVertexWithoutMargin(localNormal);
    outsupport = localNormal.dot(outpt);
  }
}

struct SweptCollector : public Collector {
  int castChild;
  SweptCollector(std::vector<ora_contact>& o, const Obj* s, World* w, int child) : Collector(o, s, w), castChild(child) {}
  virtual btScalar addSingleResult(btManifoldPoint& cp, const btCollisionObjectWrapper* o0, int p0, int i0,
                                   const btCollisionObjectWrapper* o1, int p1, int i1) {
    float rv = Collector::addSingleResult(cp, o0, p0, i0, o1, p1, i1);
    if (rv == 1) {
      bool castFirst = (o0->getCollisionObject() == self);
      btVector3 nFromCast = -(castFirst ? 1 : -1) * cp.m_normalWorldOnB;
      const SweptHull* sh = dynamic_cast<const SweptHull*>((castFirst ? o0 : o1)->getCollisionObject()->getCollisionShape());
      btTransform tw0 = self->getWorldTransform();
      btTransform tw1 = self->getWorldTransform() * sh->t01;
      btVector3 nl0 = nFromCast * tw0.getBasis();
      btVector3 nl1 = nFromCast * tw1.getBasis();
      ora_contact& col = out.back();
      const float SUPPORT_FUNC_TOLERANCE = .01;
      if (castFirst) {
        for (int k = 0; k < 3; ++k) {
          std::swap(col.ptA[k], col.ptB[k]);
          col.n[k] *= -1;
        }
        std::swap(col.objA, col.objB);
        std::swap(col.childA, col.childB);
      }
      col.childB = castChild;  // the temp object is a bare convex; report which geom of the link it was
      btVector3 pl0, pl1;
      float ls0, ls1;
      averageSupport(sh->inner, nl0, ls0, pl0);
      btVector3 pw0 = tw0 * pl0;
      averageSupport(sh->inner, nl1, ls1, pl1);
      btVector3 pw1 = tw1 * pl1;
      float sup0 = nFromCast.dot(pw0);
      float sup1 = nFromCast.dot(pw1);
      if (sup0 - sup1 > SUPPORT_FUNC_TOLERANCE) {
        col.time = 0;
        col.cctype = ORA_CC_TIME0;
      } else if (sup1 - sup0 > SUPPORT_FUNC_TOLERANCE) {
        col.time = 1;
        col.cctype = ORA_CC_TIME1;
      } else {
        const btVector3& ptOnCast = castFirst ? cp.m_positionWorldOnA : cp.m_positionWorldOnB;
        float l0c = (ptOnCast - pw0).length(), l1c = (ptOnCast - pw1).length();
        for (int k = 0; k < 3; ++k) {
          col.ptB[k] = pw0[k];
          col.ptB1[k] = pw1[k];
        }
        col.cctype = ORA_CC_BETWEEN;
        const float LENGTH_TOLERANCE = .001;
        if (l0c + l1c < LENGTH_TOLERANCE)
          col.time = .5;
        else
          col.time = l0c / (l0c + l1c);
      }
    }
    return rv;
  }
};

btCollisionShape* makeShape(const ora_scene* s, int si) {
  switch (s->shape_kind[si]) {
    case ORA_BOX:
      return new btBoxShape(btVector3(s->shape_param[4 * si], s->shape_param[4 * si + 1], s->shape_param[4 * si + 2]));
    case ORA_SPHERE:
      return new btSphereShape(s->shape_param[4 * si]);
    case ORA_CYLZ: {
      float r = s->shape_param[4 * si], h = s->shape_param[4 * si + 1];
      return new btCylinderShapeZ(btVector3(r, r, h));
    }
    case ORA_HULL: {
      btConvexHullShape* h = new btConvexHullShape();
      const float* v = s->verts + 3 * s->shape_voff[si];
      for (int i = 0; i < s->shape_vcnt[si]; ++i) h->addPoint(btVector3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
      return h;
    }
  }
  throw std::runtime_error("bad shape kind");
}

void castShape(World* w, btCollisionShape* shape, const btTransform& tf0, const btTransform& tf1, Obj* cow, int child,
               std::vector<ora_contact>& out) {
  if (btConvexShape* convex = dynamic_cast<btConvexShape*>(shape)) {
    SweptHull* sh = new SweptHull(convex, tf0.inverseTimes(tf1));
    Obj* tmp = new Obj(cow->id);
    tmp->setCollisionShape(sh);
    tmp->setWorldTransform(tf0);
    SweptCollector cc(out, tmp, w, child);
    cc.m_collisionFilterMask = ORA_GROUP_KINBODY;
    w->world->contactTest(tmp, cc);
    delete tmp;
    delete sh;
  } else if (btCompoundShape* comp = dynamic_cast<btCompoundShape*>(shape)) {
    for (int i = 0; i < comp->getNumChildShapes(); ++i)
      castShape(w, comp->getChildShape(i), tf0 * comp->getChildTransform(i), tf1 * comp->getChildTransform(i), cow, i, out);
  } else
    throw std::runtime_error("cast: convex or compound only");
}

}  // namespace

extern "C" {

void* ref_create(const ora_scene* s) {
  World* w = new World;
  w->cfg = new btDefaultCollisionConfiguration();
  w->disp = new btCollisionDispatcher(w->cfg);
  w->bp = new btDbvtBroadphase();
  w->world = new btCollisionWorld(w->disp, w->bp, w->cfg);
  w->disp->registerCollisionCreateFunc(BOX_SHAPE_PROXYTYPE, BOX_SHAPE_PROXYTYPE,
                                       w->cfg->getCollisionAlgorithmCreateFunc(CONVEX_SHAPE_PROXYTYPE, CONVEX_SHAPE_PROXYTYPE));
  w->n = s->n_objects;
  w->allowed.assign((size_t)w->n * w->n, 1);
  w->setContactDistance(.05f);
  const float MARGIN = 0;
  for (int o = 0; o < s->n_objects; ++o) {
    Obj* cow = new Obj(o);
    int c0 = s->obj_child_off[o], nc = s->obj_child_cnt[o];
    if (!s->obj_compound[o]) {
      btCollisionShape* shape = makeShape(s, s->child_shape[c0]);
      shape->setMargin(MARGIN);
      w->owned.push_back(shape);
      cow->setCollisionShape(shape);
    } else {
      btCompoundShape* comp = new btCompoundShape(false);
      w->owned.push_back(comp);
      comp->setMargin(MARGIN);
      cow->setCollisionShape(comp);
      for (int c = 0; c < nc; ++c) {
        btCollisionShape* sub = makeShape(s, s->child_shape[c0 + c]);
        w->owned.push_back(sub);
        sub->setMargin(MARGIN);
        comp->addChildShape(toBt7(s->child_T + 7 * (c0 + c)), sub);
      }
    }
    cow->setWorldTransform(btTransform::getIdentity());
    w->world->addCollisionObject(cow, (short)s->obj_group[o]);
    cow->setContactProcessingThreshold(w->contactDistance);
    w->objs.push_back(cow);
  }
  return w;
}

void ref_destroy(void* h) {
  World* w = (World*)h;
  for (size_t i = 0; i < w->objs.size(); ++i) {
    w->world->removeCollisionObject(w->objs[i]);
    delete w->objs[i];
  }
  for (size_t i = 0; i < w->owned.size(); ++i) delete w->owned[i];
  delete w->world;
  delete w->bp;
  delete w->disp;
  delete w->cfg;
  delete w;
}

void ref_set_contact_distance(void* h, float d) { ((World*)h)->setContactDistance(d); }

void ref_set_allowed(void* h, const unsigned char* m) {
  World* w = (World*)h;
  memcpy(&w->allowed[0], m, (size_t)w->n * w->n);
}

static void syncWorld(World* w, const double* objT) {
  for (int i = 0; i < w->n; ++i) w->objs[i]->setWorldTransform(toBt7(objT + 7 * i));
}

int ref_links_vs_all(void* h, const double* objT, const int* q, int nq, int filterMask, ora_contact* out, int cap) {
  World* w = (World*)h;
  std::vector<ora_contact> v;
  syncWorld(w, objT);
  w->world->updateAabbs();
  for (int i = 0; i < nq; ++i) {
    Obj* cow = w->objs[q[i]];
    Collector cc(v, cow, w);
    cc.m_collisionFilterMask = (short)filterMask;
    w->world->contactTest(cow, cc);
  }
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

int ref_cast_vs_all(void* h, const double* objT, const int* q, int nq, const double* T0, const double* T1, ora_contact* out,
                    int cap) {
  World* w = (World*)h;
  std::vector<ora_contact> v;
  syncWorld(w, objT);
  w->world->updateAabbs();
  for (int i = 0; i < nq; ++i) {
    Obj* cow = w->objs[q[i]];
    castShape(w, cow->getCollisionShape(), toBt7(T0 + 7 * i), toBt7(T1 + 7 * i), cow, 0, v);
  }
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

/* ContinuousCheckTrajectory / ContinuousCheckShape (bullet_collision_checker.cpp:683-753) through the reference's own
 * btCollisionWorld::convexSweepTest.  traj: [nsteps][nq][7] world poses of the query links. */
namespace {
struct ClosestWithChild : public btCollisionWorld::ClosestConvexResultCallback {
  int hitChild;
  ClosestWithChild() : btCollisionWorld::ClosestConvexResultCallback(btVector3(NAN, NAN, NAN), btVector3(NAN, NAN, NAN)), hitChild(0) {}
  virtual btScalar addSingleResult(btCollisionWorld::LocalConvexResult& r, bool normalInWorldSpace) {
    hitChild = r.m_localShapeInfo ? r.m_localShapeInfo->m_triangleIndex : 0;   // LocalInfoAdder: index of the compound child
    return btCollisionWorld::ClosestConvexResultCallback::addSingleResult(r, normalInWorldSpace);
  }
};
void sweepShape(World* w, btCollisionShape* shape, const std::vector<btTransform>& tfs, Obj* cow, int child, std::vector<ora_sweep_hit>& out) {
  if (btConvexShape* convex = dynamic_cast<btConvexShape*>(shape)) {
    for (int i = 0; i + 1 < (int)tfs.size(); ++i) {
      ClosestWithChild ccc;
      ccc.m_collisionFilterMask = ORA_GROUP_KINBODY;
      w->world->convexSweepTest(convex, tfs[i], tfs[i + 1], ccc, 0);
      if (ccc.hasHit()) {
        ora_sweep_hit h;
        memset(&h, 0, sizeof h);
        h.obj = cow->id; h.child = child; h.seg = i;
        h.hitObj = static_cast<const Obj*>(ccc.m_hitCollisionObject)->id;
        h.hitChild = ccc.hitChild;
        h.fraction = ccc.m_closestHitFraction;
        for (int k = 0; k < 3; ++k) { h.n[k] = ccc.m_hitNormalWorld[k]; h.pt[k] = ccc.m_hitPointWorld[k]; }
        out.push_back(h);
      }
    }
  } else if (btCompoundShape* comp = dynamic_cast<btCompoundShape*>(shape)) {
    for (int i = 0; i < comp->getNumChildShapes(); ++i) {
      std::vector<btTransform> t2(tfs.size());
      for (size_t k = 0; k < tfs.size(); ++k) t2[k] = tfs[k] * comp->getChildTransform(i);
      sweepShape(w, comp->getChildShape(i), t2, cow, i, out);
    }
  } else
    throw std::runtime_error("sweep: convex or compound only");
}
}  // namespace

int ref_continuous_check(void* h, const double* objT, const int* q, int nq, const double* traj, int nsteps, ora_sweep_hit* out, int cap) {
  World* w = (World*)h;
  std::vector<ora_sweep_hit> v;
  syncWorld(w, objT);
  w->world->updateAabbs();
  for (int i = 0; i < nq; ++i) {
    Obj* cow = w->objs[q[i]];
    std::vector<btTransform> tfs(nsteps);
    for (int s = 0; s < nsteps; ++s) tfs[s] = toBt7(traj + 7 * ((size_t)s * nq + i));
    sweepShape(w, cow->getCollisionShape(), tfs, cow, 0, v);
  }
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
  return n;
}

/* RayCastCollision (bullet_collision_checker.cpp:352-360) through the reference's own btCollisionWorld::rayTest;
 * closest[i] = the callback's m_closestHitFraction (1 = no hit). */
int ref_ray_cast(void* h, const double* objT, const double* from3, const double* to3, int n_rays, float* closest) {
  World* w = (World*)h;
  syncWorld(w, objT);
  w->world->updateAabbs();
  int hits = 0;
  for (int i = 0; i < n_rays; ++i) {
    btVector3 btFrom((float)from3[3 * i], (float)from3[3 * i + 1], (float)from3[3 * i + 2]);
    btVector3 btTo((float)to3[3 * i], (float)to3[3 * i + 1], (float)to3[3 * i + 2]);
    btCollisionWorld::ClosestRayResultCallback res(btFrom, btTo);
    w->world->rayTest(btFrom, btTo, res);
    closest[i] = res.m_closestHitFraction;
    hits += res.hasHit() ? 1 : 0;
  }
  return hits;
}

/* AllVsAll (bullet_collision_checker.cpp:463-498) with its near callback (:415-422): performDiscreteCollisionDetection,
 * then every point of every persistent manifold with weight 1 / numContacts; manifolds cleared afterwards.  childA / childB
 * = the manifold point's m_index0 / m_index1 (the compound algorithm's shape identifiers; -1 for a bare shape). */
static void refNearCallback(btBroadphasePair& pair, btCollisionDispatcher& dispatcher, const btDispatcherInfo& info) {
  World* w = static_cast<World*>(dispatcher.m_userData);
  if (w->canCollide(static_cast<Obj*>(pair.m_pProxy0->m_clientObject)->id, static_cast<Obj*>(pair.m_pProxy1->m_clientObject)->id))
    dispatcher.defaultNearCallback(pair, dispatcher, info);
}
int ref_all_vs_all(void* h, const double* objT, ora_contact* out, double* weight, int cap) {
  World* w = (World*)h;
  syncWorld(w, objT);
  w->disp->setNearCallback(&refNearCallback);
  w->disp->m_userData = w;
  w->world->performDiscreteCollisionDetection();
  int n = 0;
  for (int i = 0; i < w->disp->getNumManifolds(); ++i) {
    btPersistentManifold* m = w->disp->getManifoldByIndexInternal(i);
    const int nc = m->getNumContacts();
    const Obj* A = static_cast<const Obj*>(m->getBody0());
    const Obj* B = static_cast<const Obj*>(m->getBody1());
    for (int j = 0; j < nc; ++j) {
      btManifoldPoint& pt = m->getContactPoint(j);
      if (w->canCollide(A->id, B->id)) {
        if (n < cap) {
          ora_contact c;
          memset(&c, 0, sizeof c);
          c.objA = A->id, c.objB = B->id, c.childA = pt.m_index0, c.childB = pt.m_index1;
          for (int k = 0; k < 3; ++k) c.ptA[k] = pt.getPositionWorldOnA()[k], c.ptB[k] = pt.getPositionWorldOnB()[k], c.n[k] = pt.m_normalWorldOnB[k];
          c.dist = pt.m_distance1;
          out[n] = c;
          weight[n] = 1. / nc;
        }
        ++n;
      }
    }
    m->clearManifold();
  }
  return n;
}

/* Post-construction vertex data of a shape, for hull shapes built elsewhere (diagnostics). */
int ref_gjk_checks(void) { return gNumGjkChecks; }
int ref_epa_checks(void) { return gNumDeepPenetrationChecks; }

/* trimesh -> hull vertices, as createShapePrimitive does for meshes with >= 50 vertices
 * (bullet_collision_checker.cpp:174-203): btShapeHull over a btConvexTriangleMeshShape. */
int ref_shape_hull(const float* verts, int nverts, const int* tri, int ntri, float* out, int cap) {
  btTriangleMesh mesh;
  for (int i = 0; i < ntri; ++i) {
    const float *a = verts + 3 * tri[3 * i], *b = verts + 3 * tri[3 * i + 1], *c = verts + 3 * tri[3 * i + 2];
    mesh.addTriangle(btVector3(a[0], a[1], a[2]), btVector3(b[0], b[1], b[2]), btVector3(c[0], c[1], c[2]));
  }
  btConvexTriangleMeshShape cts(&mesh);
  cts.setMargin(0);
  btShapeHull sh(&cts);
  if (nverts >= 50 && sh.buildHull(-666)) {
    int n = sh.numVertices();
    for (int i = 0; i < n && i < cap; ++i)
      for (int k = 0; k < 3; ++k) out[3 * i + k] = sh.getVertexPointer()[i][k];
    return n;
  }
  for (int i = 0; i < nverts && i < cap; ++i)
    for (int k = 0; k < 3; ++k) out[3 * i + k] = verts[3 * i + k];
  return nverts;
}

}  // extern "C"
