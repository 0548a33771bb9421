"""Host-side mirror of the reference interface for this path, on top of the C-ABI engine.

Same names, argument meaning and error behaviour as the reference (OpenRAVE types replaced by the flat Scene /
RobotModel of model.py, because OpenRAVE is not available here):

  Collision, CastCollisionType            src/trajopt/collision_checker.hpp:12-29
  HIPCollisionChecker(CollisionChecker)   src/trajopt/collision_checker.hpp:40-84, bullet_collision_checker.cpp:321-389
  CreateCollisionChecker                  src/trajopt/bullet_collision_checker.cpp:1036-1039
  SingleTimestep/CastCollisionEvaluator   src/trajopt/collision_terms.cpp:82-174
  CollisionCost                           src/trajopt/collision_terms.cpp:196-223

The one architectural addition is batching: the reference evaluates one timestep per checker call; the
evaluators here are fed by TrajectoryLinearizer, which sends the whole trajectory (and many independent problems) to
the GPU in one call and serves the per-timestep evaluators from that slab, keyed like the reference's 3-entry cache
(collision_terms.cpp:82-98) by the DOF sub-vector.
"""
import numpy as np

from . import engine as _engine
from . import model as M

CCType_None, CCType_Time0, CCType_Time1, CCType_Between = 0, 1, 2, 3
RobotFilter, KinBodyFilter = 1, 2


class Collision:
    """collision_checker.hpp:19-29.  linkA/linkB are object ids (the C++ adapter maps them to KinBody::Link*)."""
    __slots__ = ("linkA", "linkB", "ptA", "ptB", "normalB2A", "ptB1", "distance", "weight", "time", "cctype",
                 "childA", "childB", "toi")

    def __init__(self, linkA, linkB, ptA, ptB, normalB2A, distance, weight=1.0, time=0.0):
        self.linkA, self.linkB = linkA, linkB
        self.ptA, self.ptB, self.normalB2A = np.asarray(ptA), np.asarray(ptB), np.asarray(normalB2A)
        self.ptB1 = np.zeros(3)
        self.distance, self.weight = float(distance), float(weight)
        self.time = 0.0           # the reference's constructor ignores its `time` argument (collision_checker.hpp:28)
        self.cctype = CCType_None
        self.childA = self.childB = 0

    def __repr__(self):
        return "%s--%s distance: %g normal: %s ptA: %s ptB: %s time: %g weight: %g" % (
            self.linkA, self.linkB, self.distance, self.normalB2A, self.ptA, self.ptB, self.time, self.weight)


def _collisions_from_records(rec):
    out = []
    for r in rec:
        c = Collision(int(r["idA"]), int(r["idB"]), r["ptA"].copy(), r["ptB"].copy(), r["n"].copy(), r["dist"])
        c.ptB1 = r["ptB1"].copy()
        c.time = float(r["time"])
        c.cctype = int(r["cctype"])
        c.childA, c.childB = int(r["childA"]), int(r["childB"])
        out.append(c)
    return out


class HIPCollisionChecker:
    """Implements trajopt::CollisionChecker on the MI355X engine.  One checker per environment; it owns a copy of the
    world (collision_checker.hpp:37-39).  Poses of non-robot bodies are `static_T`; the robot's pose is whatever
    SetDOFValues last set (the reference reads it from the OpenRAVE environment, bullet_collision_checker.cpp:648-653)."""

    def __init__(self, scene, static_T, robot=None, device=0):
        self.scene, self.robot = scene, robot
        self.static_T = np.ascontiguousarray(static_T, np.float64).reshape(-1, 7).copy()
        self.eng = _engine.Engine(device)     # raises when the HIP library or a GPU is missing: no fallback
        self.eng.upload_scene(scene, self.static_T)
        self.allowed = np.ones((scene.n_objects, scene.n_objects), np.uint8)
        self._excluded = set()
        self.SetContactDistance(.05)          # bullet_collision_checker.cpp:437
        self._dofs = None
        if robot is not None:
            self.SetRobot(robot)

    # ---- set-up ---------------------------------------------------------------------------------------
    def SetRobot(self, robot, ignore_zero_state=True):
        """AddKinBody for a robot + IgnoreZeroStateSelfCollisions at d = 0.1 (bullet_collision_checker.cpp:629-643)."""
        self.robot = robot
        self.eng.upload_robot(robot)
        self._dofs = np.zeros(robot.K)
        if ignore_zero_state:
            old = self.GetContactDistance()
            self.SetContactDistance(.1)
            self.IgnoreZeroStateSelfCollisions()
            self.SetContactDistance(old)

    def SetContactDistance(self, distance):
        self.eng.set_contact_distance(distance)

    def GetContactDistance(self):
        return self.eng.get_contact_distance()

    def ExcludeCollisionPair(self, link0, link1):
        self._excluded.add((link0, link1))
        self.allowed[link0, link1] = 0       # one direction (bullet_collision_checker.cpp:342-346); UpdateAllowedCollisionMatrix
        self.eng.set_allowed_matrix(self.allowed)   # (:616-627, on the next body change) clears both

    def IncludeCollisionPair(self, link0, link1):
        self._excluded.discard((link0, link1))
        self.allowed[link0, link1] = 1
        self.eng.set_allowed_matrix(self.allowed)

    def IgnoreZeroStateSelfCollisions(self):
        """collision_checker.cpp:41-56: all joints zero, body lifted to z = 10 so that only self contacts show up; every
        colliding link pair is excluded for good."""
        rb = self.robot
        saved = (np.array(rb.arrays["q_robot"]), np.array(rb.arrays["base_T"]), list(rb.joint_inds), rb.affine)
        zero = M.RobotModel.__new__(M.RobotModel)
        zero.__dict__.update(rb.__dict__)
        zero.q_robot = np.zeros(max(rb.n_dof, 1))
        zero.base_T = (1, 0, 0, 0, 0, 0, 10)
        zero.joint_inds, zero.affine = list(range(rb.n_dof)) or [], 0
        if not zero.joint_inds:              # e.g. boxbot: a single free link has no self collisions
            return
        zero.finalize()
        self.eng.upload_robot(zero)
        # BodyVsAll: EVERY link of the body is a query, not only those a joint moves (collision_checker.cpp:49)
        T = self.static_T.copy()
        fk = self.eng.forward_kinematics(np.zeros((1, zero.K)))[0]
        # (a grabbed body is not part of the robot's KinBody: it is neither lifted nor a query here -- the reference runs
        # this when a body is ADDED, bullet_collision_checker.cpp:636-641, before anything can be grabbed)
        own = [i for i, o in enumerate(rb.link_obj) if o >= 0 and rb.dof[i] != M.DOF_GRABBED]
        body = [rb.link_obj[i] for i in own]
        for i in own:
            T[rb.link_obj[i]] = fk[i]
        rec = self.eng.links_vs_all_poses(T, body, -1)
        for a, b in zip(rec["idA"].tolist(), rec["idB"].tolist()):
            self.allowed[a, b] = self.allowed[b, a] = 0
            self._excluded.add((a, b))
        self.eng.set_allowed_matrix(self.allowed)
        rb.q_robot, rb.base_T = saved[0], tuple(saved[1])
        rb.joint_inds, rb.affine = saved[2], saved[3]
        rb.finalize()
        self.eng.upload_robot(rb)

    # ---- queries ----------------------------------------------------------------------------------------
    def SetDOFValues(self, dofs):
        """Configuration::SetDOFValues -- the pose LinksVsAll will see."""
        self._dofs = np.asarray(dofs, float).copy()

    def LinksVsAll(self, links, collisions, filterMask=-1):
        """Appends to `collisions` (never clears it, bullet_collision_checker.cpp:404).  `links`: object ids."""
        if self.robot is None:
            rec = self.eng.links_vs_all_poses(self.static_T, list(links), filterMask)
        else:
            rec, _ = self.eng.links_vs_all_batch(self._dofs[None], 1, 1, filterMask, want_jacobians=False)
            want = set(int(x) for x in links)
            q = set(self.robot.affected_links()[0])
            if want != q:
                # a query that is not "the affected links": answer it on explicit poses
                T = self._current_poses()
                rec = self.eng.links_vs_all_poses(T, list(links), filterMask)
        collisions.extend(_collisions_from_records(rec))

    def LinkVsAll(self, link, collisions, filterMask=-1):
        self.LinksVsAll([link], collisions, filterMask)

    def BodyVsAll(self, body_links, collisions, filterMask=-1):
        self.LinksVsAll(body_links, collisions, filterMask)

    def AllVsAll(self, collisions):
        """bullet_collision_checker.cpp:463-498 ("seems to be broken" upstream): the points of Bullet's persistent manifolds
        after performDiscreteCollisionDetection = every unordered link pair once, one contact per convex pair up to the
        breaking threshold 2.001 d (not d), weight 1/numContacts = 1 (tohc_all_vs_all_poses)."""
        rec = self.eng.all_vs_all_poses(self._current_poses())
        collisions.extend(_collisions_from_records(rec))   # Collision.weight defaults to 1

    def CastVsAll(self, rad, links, startjoints, endjoints, collisions):
        dofs = np.stack([np.asarray(startjoints, float), np.asarray(endjoints, float)])
        rec, _ = self.eng.cast_vs_all_batch(dofs, 1, 2, 1, want_jacobians=False)
        collisions.extend(_collisions_from_records(rec))

    def ContinuousCheckTrajectory(self, traj, rad, collisions):
        """bullet_collision_checker.cpp:708-753: per affected link geometry and trajectory segment the first object
        hit on the way (conservative advancement on the GPU).  Appends Collision(link, hitLink, hitPoint, hitPoint,
        hitNormal, 0, 1, seg + fraction); like the reference's constructor the Collision keeps time = 0
        (collision_checker.hpp:28), so the time of impact is also stored as `.toi`."""
        traj = np.ascontiguousarray(traj, float).reshape(-1, self.robot.K)
        rec = self.eng.continuous_check_trajectory(traj, 1, len(traj))
        # the reference's order: link, geometry, segment (ContinuousCheckShape recursion, :683-705)
        order = np.lexsort((rec["t"], rec["childA"], rec["idA"]))
        for r in rec[order]:
            c = Collision(int(r["idA"]), int(r["idB"]), r["ptA"].copy(), r["ptB"].copy(), r["n"].copy(), 0.0, 1.0,
                          float(r["t"]) + float(r["time"]))
            c.childA, c.childB = int(r["childA"]), int(r["childB"])
            c.toi = float(np.float32(r["t"]) + np.float32(r["time"]))
            collisions.append(c)

    def RayCastCollision(self, point1, point2):
        """bullet_collision_checker.cpp:352-360: does the segment point1 -> point2 hit anything (robot links included)
        at the current configuration?"""
        return bool(self.RayCastCollisions([point1], [point2])[0] < 1.0)

    def RayCastCollisions(self, points1, points2):
        """Batched form (no counterpart in the reference, which takes one ray per call): closest hit fraction per ray,
        1 = nothing hit."""
        if self.robot is None:
            return self.eng.ray_cast_poses(self.static_T, points1, points2)
        return self.eng.ray_cast(self._dofs, points1, points2)

    def _current_poses(self):
        T = self.static_T.copy()
        if self.robot is not None:
            fk = self.eng.forward_kinematics(self._dofs[None])[0]
            for i, o in enumerate(self.robot.link_obj):
                if o >= 0:
                    T[o] = fk[i]
        return T


def CreateCollisionChecker(scene, static_T, robot=None, device=0):
    return HIPCollisionChecker(scene, static_T, robot, device)


# ----------------------------------------------------------------------------------------------------------------
# batched linearization + the reference's evaluators / cost on top of it
# ----------------------------------------------------------------------------------------------------------------
class TrajectoryLinearizer:
    """The "prefetch" of SURVEY.md 7(8): one GPU call per trajectory batch, results served per timestep."""

    def __init__(self, checker):
        self.cc = checker
        self.K = checker.robot.K

    def linearize(self, dofs, cast=False, gap=1, filter_mask=-1, want_jacobians=True):
        """dofs: [B][T][K] -> (records, grads) in canonical order."""
        dofs = np.ascontiguousarray(dofs, float)
        if dofs.ndim == 2:
            dofs = dofs[None]
        B, T, _ = dofs.shape
        if cast:
            return self.cc.eng.cast_vs_all_batch(dofs, B, T, gap, want_jacobians)
        return self.cc.eng.links_vs_all_batch(dofs, B, T, filter_mask, want_jacobians)


def aff_exprs(rec, grads, cast):
    """CollisionsToDistanceExpressions (collision_terms.cpp:37-80): per contact (constant, coeffs0, coeffs1)."""
    if not cast:
        return rec["const0"].copy(), grads[:, 0].copy(), None
    t = rec["time"].astype(float)
    one_minus_t = (np.float32(1) - rec["time"]).astype(float)    # `1-collisions[i].time` is float arithmetic (:73)
    const = one_minus_t * rec["const0"] + t * rec["const1"]
    return const, one_minus_t[:, None] * grads[:, 0], t[:, None] * grads[:, 1]


def bpmpd_collision_block(rec, grads, B, T, K, gap, dist_pen_discrete, dist_pen_cast):
    """What sco hands BPMPD for the collision hinges of one linearization, built on the HOST from the contact slab the
    way the reference does it -- the checker of tohc_fetch_hinges:
      CollisionsToDistanceExpressions (collision_terms.cpp:37-80) -> CollisionCost::convex (:205-214: viol = dist_pen - expr)
      -> ConvexObjective::addHinge (modeling.cpp:22-29: row viol - h <= 0) -> BPMPDModel::optimize (bpmpd_interface.cpp:352-369:
      rhs = -constant, per variable the (row, value) list through simplify2: zeros dropped, rows ascending).
    rec / grads: one slab in canonical order.  Returns the dict Engine.fetch_hinges returns."""
    n = len(rec)
    cast = rec["cctype"] != 0
    const = rec["const0"].copy()
    wa, wb = np.ones(n), np.zeros(n)
    wa[cast] = (np.float32(1) - rec["time"][cast]).astype(float)      # `1-collisions[i].time` is float arithmetic (:73)
    wb[cast] = rec["time"][cast].astype(float)
    const[cast] = wa[cast] * rec["const0"][cast] + wb[cast] * rec["const1"][cast]
    pen = np.where(cast, dist_pen_cast, dist_pen_discrete)
    rhs = -(pen - const)
    row_off = np.searchsorted(rec["problem"], np.arange(B + 1)).astype(np.int64)
    cols = [[] for _ in range(B * T * K)]
    for i in range(n):                                                # rows in canonical order = ascending within a column
        p, t = int(rec["problem"][i]), int(rec["t"][i])
        r = i - int(row_off[p])
        v0 = -(wa[i] * grads[i, 0])
        for k in np.nonzero(v0 != 0)[0]:
            cols[(p * T + t) * K + k].append((r, v0[k]))
        if cast[i]:
            v1 = -(wb[i] * grads[i, 1])
            for k in np.nonzero(v1 != 0)[0]:
                cols[(p * T + t + gap) * K + k].append((r, v1[k]))
    for c in cols:
        c.sort(key=lambda e: e[0])                                    # simplify2's std::map order (no row repeats within a column)
    col_off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    col_idx = np.array([e[0] for c in cols for e in c], np.int32)
    col_val = np.array([e[1] for c in cols for e in c], np.float64)
    from .engine import HINGE_CAST
    return dict(B=B, T=T, K=K, gap=gap, n_rows=n, nnz=len(col_val), row_off=row_off, rhs=rhs, dist=rec["dist"].copy(),
                row_t=(rec["t"] | np.where(cast, HINGE_CAST, 0)).astype(np.int32), col_off=col_off, col_idx=col_idx, col_val=col_val)


class CollisionCost:
    """collision_terms.cpp:196-223: hinge cost coeff * max(0, dist_pen - dist) per contact."""

    def __init__(self, dist_pen, coeff, linearizer, cast=False, gap=1):
        self.dist_pen, self.coeff, self.lin, self.cast, self.gap = dist_pen, coeff, linearizer, cast, gap

    def value(self, traj):
        rec, _ = self.lin.linearize(traj, self.cast, self.gap, want_jacobians=False)
        return float(np.maximum(0.0, self.dist_pen - rec["dist"]).sum() * self.coeff)

    def convex(self, traj):
        """-> list of (t, constant, coeffs0, coeffs1): hinge(dist_pen - (constant + coeffs . x)) * coeff."""
        rec, g = self.lin.linearize(traj, self.cast, self.gap)
        const, c0, c1 = aff_exprs(rec, g, self.cast)
        return [(int(rec["t"][i]), self.dist_pen - const[i], -c0[i], None if c1 is None else -c1[i]) for i in range(len(rec))]
