"""ctypes bindings for the two CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY, see oracle_api.h).

Importers allowed: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
  * ``load_oracle()`` -> oracle/liboracle.so   (CPU restatement; built by ``make -C oracle oracle``)
  * ``load_ref()``    -> oracle/_ref/libref.so (reference's vendored Bullet; only where it was built)
Both expose the same world API over the flat ``ora_scene`` of oracle_api.h; scenes and robots are described with
trajopt_amd.model.Scene / RobotModel (plain numpy arrays).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OraScene(C.Structure):
    _fields_ = [
        ("n_shapes", C.c_int), ("shape_kind", C.POINTER(C.c_int)), ("shape_param", C.POINTER(C.c_float)),
        ("shape_voff", C.POINTER(C.c_int)), ("shape_vcnt", C.POINTER(C.c_int)),
        ("n_verts", C.c_int), ("verts", C.POINTER(C.c_float)),
        ("n_children", C.c_int), ("child_shape", C.POINTER(C.c_int)), ("child_T", C.POINTER(C.c_double)),
        ("n_objects", C.c_int), ("obj_child_off", C.POINTER(C.c_int)), ("obj_child_cnt", C.POINTER(C.c_int)),
        ("obj_compound", C.POINTER(C.c_int)), ("obj_group", C.POINTER(C.c_int)),
    ]


class OraContact(C.Structure):
    _fields_ = [
        ("objA", C.c_int), ("childA", C.c_int), ("objB", C.c_int), ("childB", C.c_int),
        ("cctype", C.c_int), ("time", C.c_float), ("dist", C.c_double),
        ("n", C.c_double * 3), ("ptA", C.c_double * 3), ("ptB", C.c_double * 3), ("ptB1", C.c_double * 3),
    ]


class OraRobot(C.Structure):
    _fields_ = [
        ("n_links", C.c_int), ("parent", C.POINTER(C.c_int)), ("jtype", C.POINTER(C.c_int)),
        ("T_origin", C.POINTER(C.c_double)), ("axis", C.POINTER(C.c_double)), ("dof", C.POINTER(C.c_int)),
        ("mult", C.POINTER(C.c_double)), ("offs", C.POINTER(C.c_double)), ("link_obj", C.POINTER(C.c_int)),
        ("n_dof", C.c_int), ("q_robot", C.POINTER(C.c_double)), ("base_T", C.c_double * 7),
        ("n_joint_inds", C.c_int), ("joint_inds", C.POINTER(C.c_int)), ("affine", C.c_int),
        ("rot_axis", C.c_double * 3),
    ]


class OraLin(C.Structure):
    _fields_ = [("t", C.c_int), ("c", OraContact), ("const0", C.c_double), ("const1", C.c_double)]


class OraSweepHit(C.Structure):
    _fields_ = [("obj", C.c_int), ("child", C.c_int), ("seg", C.c_int), ("hitObj", C.c_int), ("hitChild", C.c_int),
                ("fraction", C.c_float), ("n", C.c_double * 3), ("pt", C.c_double * 3)]


SWEEP_DTYPE = np.dtype([("obj", "i4"), ("child", "i4"), ("seg", "i4"), ("hitObj", "i4"), ("hitChild", "i4"),
                        ("fraction", "f4"), ("n", "f8", 3), ("pt", "f8", 3)], align=True)
assert SWEEP_DTYPE.itemsize == C.sizeof(OraSweepHit)

CONTACT_DTYPE = np.dtype([
    ("objA", "i4"), ("childA", "i4"), ("objB", "i4"), ("childB", "i4"), ("cctype", "i4"), ("time", "f4"),
    ("dist", "f8"), ("n", "f8", 3), ("ptA", "f8", 3), ("ptB", "f8", 3), ("ptB1", "f8", 3)], align=True)
assert CONTACT_DTYPE.itemsize == C.sizeof(OraContact)
LIN_DTYPE = np.dtype([("t", "i4"), ("c", CONTACT_DTYPE), ("const0", "f8"), ("const1", "f8")], align=True)
assert LIN_DTYPE.itemsize == C.sizeof(OraLin)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def make_c_scene(scene):
    """ctypes view of a finalized trajopt_amd.model.Scene (arrays stay alive through scene.arrays)."""
    a = scene.arrays
    s = OraScene()
    s.n_shapes = len(a["kind"]); s.shape_kind = _p(a["kind"], C.c_int); s.shape_param = _p(a["param"], C.c_float)
    s.shape_voff = _p(a["voff"], C.c_int); s.shape_vcnt = _p(a["vcnt"], C.c_int)
    s.n_verts = int(a["vcnt"].sum()); s.verts = _p(a["verts"], C.c_float)
    s.n_children = len(a["child_shape"]); s.child_shape = _p(a["child_shape"], C.c_int)
    s.child_T = _p(a["child_T"], C.c_double)
    s.n_objects = len(a["obj_off"]); s.obj_child_off = _p(a["obj_off"], C.c_int)
    s.obj_child_cnt = _p(a["obj_cnt"], C.c_int); s.obj_compound = _p(a["obj_compound"], C.c_int)
    s.obj_group = _p(a["obj_group"], C.c_int)
    return s


def make_c_robot(robot):
    """ctypes view of a finalized trajopt_amd.model.RobotModel."""
    a = robot.arrays
    r = OraRobot()
    r.n_links = robot.n_links
    r.parent = _p(a["parent"], C.c_int); r.jtype = _p(a["jtype"], C.c_int)
    r.T_origin = _p(a["T_origin"], C.c_double); r.axis = _p(a["axis"], C.c_double)
    r.dof = _p(a["dof"], C.c_int); r.mult = _p(a["mult"], C.c_double); r.offs = _p(a["offs"], C.c_double)
    r.link_obj = _p(a["link_obj"], C.c_int)
    r.n_dof = robot.n_dof; r.q_robot = _p(a["q_robot"], C.c_double)
    for i in range(7):
        r.base_T[i] = float(a["base_T"][i])
    r.n_joint_inds = len(robot.joint_inds); r.joint_inds = _p(a["joint_inds"], C.c_int)
    r.affine = robot.affine
    for i in range(3):
        r.rot_axis[i] = float(a["rot_axis"][i])
    return r


class World:
    """One checker world (either library)."""

    def __init__(self, lib, prefix, scene, cap=1 << 16):
        self.lib, self.px, self.scene = lib, prefix, scene
        f = lambda n: getattr(lib, prefix + n)
        f("create").restype = C.c_void_p
        f("create").argtypes = [C.POINTER(OraScene)]
        f("destroy").argtypes = [C.c_void_p]
        f("set_contact_distance").argtypes = [C.c_void_p, C.c_float]
        f("set_allowed").argtypes = [C.c_void_p, C.POINTER(C.c_ubyte)]
        f("links_vs_all").restype = C.c_int
        f("links_vs_all").argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_int,
                                      C.POINTER(OraContact), C.c_int]
        f("cast_vs_all").restype = C.c_int
        f("cast_vs_all").argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OraContact), C.c_int]
        if hasattr(lib, prefix + "continuous_check"):   # the host emulation of the device code has it too
            f("continuous_check").restype = C.c_int
            f("continuous_check").argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                                              C.POINTER(C.c_double), C.c_int, C.POINTER(OraSweepHit), C.c_int]
        self.f = f
        self.cscene = make_c_scene(scene)
        self.h = f("create")(C.byref(self.cscene))
        self.cap = cap
        self.buf = np.zeros(cap, CONTACT_DTYPE)
        if prefix == "ora_":
            lib.ora_counter.restype = C.c_long
            lib.ora_counter.argtypes = [C.c_void_p, C.c_int]
            for n in ("linearize_discrete", "linearize_cast"):
                f(n).restype = C.c_int
            lib.ora_linearize_discrete.argtypes = [C.c_void_p, C.POINTER(OraRobot), C.POINTER(C.c_double),
                                                   C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(OraLin),
                                                   C.POINTER(C.c_double), C.c_int]
            lib.ora_linearize_cast.argtypes = [C.c_void_p, C.POINTER(OraRobot), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(OraLin),
                                               C.POINTER(C.c_double), C.c_int]

    def close(self):
        if self.h:
            self.f("destroy")(self.h)
            self.h = None

    def set_contact_distance(self, d):
        self.f("set_contact_distance")(self.h, float(d))

    def set_allowed(self, m):
        m = np.ascontiguousarray(m, dtype=np.uint8)
        assert m.shape == (self.scene.n_objects,) * 2
        self.f("set_allowed")(self.h, _p(m, C.c_ubyte))

    def links_vs_all(self, objT, query, mask=-1):
        objT = np.ascontiguousarray(objT, np.float64).reshape(-1, 7)
        assert len(objT) == self.scene.n_objects
        q = np.ascontiguousarray(query, np.int32)
        n = self.f("links_vs_all")(self.h, _p(objT, C.c_double), _p(q, C.c_int), len(q), int(mask),
                                   self.buf.ctypes.data_as(C.POINTER(OraContact)), self.cap)
        assert n <= self.cap
        return self.buf[:n].copy()

    def all_vs_all(self, objT):
        """AllVsAll (bullet_collision_checker.cpp:463-498); the reference driver also returns 1 / numContacts per point, which
        this asserts to be 1.  The reference's manifolds keep the breaking threshold they were created under: set the
        contact distance before the first call."""
        objT = np.ascontiguousarray(objT, np.float64)
        out = np.zeros(self.cap, CONTACT_DTYPE)
        f = self.f("all_vs_all")
        f.restype = C.c_int
        if self.px == "ref_":
            wt = np.zeros(self.cap)
            f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.c_int]
            n = f(self.h, objT.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.c_void_p), wt.ctypes.data_as(C.POINTER(C.c_double)), self.cap)
            assert n <= self.cap and (wt[:n] == 1.0).all()
        else:
            f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_int]
            n = f(self.h, objT.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.c_void_p), self.cap)
            assert n <= self.cap
        return out[:n].copy()

    def cast_vs_all(self, objT, query, T0, T1):
        objT = np.ascontiguousarray(objT, np.float64).reshape(-1, 7)
        q = np.ascontiguousarray(query, np.int32)
        T0 = np.ascontiguousarray(T0, np.float64).reshape(-1, 7)
        T1 = np.ascontiguousarray(T1, np.float64).reshape(-1, 7)
        assert len(T0) == len(q) == len(T1)
        n = self.f("cast_vs_all")(self.h, _p(objT, C.c_double), _p(q, C.c_int), len(q), _p(T0, C.c_double),
                                  _p(T1, C.c_double), self.buf.ctypes.data_as(C.POINTER(OraContact)), self.cap)
        assert n <= self.cap
        return self.buf[:n].copy()

    def set_trig_mode(self, portable):
        """oracle / emu: 0 = libm's sinf/cosf/acosf like the reference, 1 = the portable routines the device runs"""
        self.f("set_trig_mode")(int(portable))

    def continuous_check(self, objT, query, traj):
        """ContinuousCheckTrajectory on explicit poses: traj [nsteps][len(query)][7] -> SWEEP_DTYPE[n], ordered
        (link, geometry, segment) like the reference's loops."""
        objT = np.ascontiguousarray(objT, np.float64).reshape(-1, 7)
        q = np.ascontiguousarray(query, np.int32)
        traj = np.ascontiguousarray(traj, np.float64).reshape(-1, len(q), 7)
        out = np.zeros(self.cap, SWEEP_DTYPE)
        n = self.f("continuous_check")(self.h, _p(objT, C.c_double), _p(q, C.c_int), len(q), _p(traj, C.c_double),
                                       len(traj), out.ctypes.data_as(C.POINTER(OraSweepHit)), self.cap)
        assert n <= self.cap
        return out[:n].copy()

    def ray_cast(self, objT, p_from, p_to):
        """RayCastCollision for a batch of rays -> float32[n] closest hit fractions (1 = no hit; hit <=> < 1)"""
        fn = self.f("ray_cast")
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_float)]
        objT = np.ascontiguousarray(objT, np.float64).reshape(-1, 7)
        assert len(objT) == self.scene.n_objects
        a = np.ascontiguousarray(p_from, np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(p_to, np.float64).reshape(-1, 3)
        assert len(a) == len(b)
        out = np.ones(len(a), np.float32)
        fn(self.h, _p(objT, C.c_double), _p(a, C.c_double), _p(b, C.c_double), len(a), _p(out, C.c_float))
        return out

    def sweep_pair(self, Q, qc, T0, T1, X, xc, XT):
        """oracle only: time of impact of ONE (swept geometry, resting geometry) pair -> SWEEP_DTYPE record or None"""
        self.lib.ora_sweep_pair.restype = C.c_int
        self.lib.ora_sweep_pair.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                            C.c_int, C.POINTER(C.c_double), C.POINTER(OraSweepHit)]
        out = np.zeros(1, SWEEP_DTYPE)
        a = [np.ascontiguousarray(t, np.float64).reshape(7) for t in (T0, T1, XT)]
        ok = self.lib.ora_sweep_pair(self.h, int(Q), int(qc), _p(a[0], C.c_double), _p(a[1], C.c_double), int(X), int(xc),
                                     _p(a[2], C.c_double), out.ctypes.data_as(C.POINTER(OraSweepHit)))
        return out[0] if ok else None

    # -- oracle only: full linearization (FK + contacts + Jacobian projection) ---------------------------
    def counter(self, which):
        return int(self.lib.ora_counter(self.h, which))

    def linearize(self, robot, static_T, dofs, cast=False, gap=1, mask=-1):
        """-> (records LIN_DTYPE[n], grads float64[n,2,K])"""
        crobot = make_c_robot(robot)
        K = robot.K
        dofs = np.ascontiguousarray(dofs, np.float64).reshape(-1, K)
        static_T = np.ascontiguousarray(static_T, np.float64).reshape(-1, 7)
        assert len(static_T) == self.scene.n_objects
        out = np.zeros(self.cap, LIN_DTYPE)
        grads = np.zeros((self.cap, 2, K))
        if cast:
            n = self.lib.ora_linearize_cast(self.h, C.byref(crobot), _p(static_T, C.c_double), _p(dofs, C.c_double),
                                            len(dofs), gap, out.ctypes.data_as(C.POINTER(OraLin)),
                                            _p(grads, C.c_double), self.cap)
        else:
            n = self.lib.ora_linearize_discrete(self.h, C.byref(crobot), _p(static_T, C.c_double),
                                                _p(dofs, C.c_double), len(dofs), int(mask),
                                                out.ctypes.data_as(C.POINTER(OraLin)), _p(grads, C.c_double), self.cap)
        assert n <= self.cap, "oracle result buffer too small"
        return out[:n].copy(), grads[:n].copy()


def oracle_fk(robot, dofs):
    lib = load_oracle()
    lib.ora_fk.argtypes = [C.POINTER(OraRobot), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cr = make_c_robot(robot)
    dofs = np.ascontiguousarray(dofs, np.float64)
    out = np.zeros((robot.n_links, 7))
    lib.ora_fk(C.byref(cr), _p(dofs, C.c_double), _p(out, C.c_double))
    return out


def oracle_position_jacobian(robot, dofs, link, pt):
    lib = load_oracle()
    lib.ora_position_jacobian.argtypes = [C.POINTER(OraRobot), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cr = make_c_robot(robot)
    dofs = np.ascontiguousarray(dofs, np.float64)
    linkT = oracle_fk(robot, dofs)
    pt = np.ascontiguousarray(pt, np.float64)
    J = np.zeros((3, robot.K))
    lib.ora_position_jacobian(C.byref(cr), _p(linkT, C.c_double), _p(dofs, C.c_double), int(link), _p(pt, C.c_double),
                              _p(J, C.c_double))
    return J


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def load_ref():
    return C.CDLL(os.path.join(HERE, "_ref", "libref.so"))


def load_oracle():
    return C.CDLL(os.path.join(HERE, "liboracle.so"))


def ref_shape_hull(vertices, triangles):
    """The reference's own btShapeHull over a btConvexTriangleMeshShape (ref_driver.cpp: ref_shape_hull), as
    createShapePrimitive calls it (bullet_collision_checker.cpp:174-203)."""
    lib = load_ref()
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
    cap = max(len(v), 64)
    out = np.zeros((cap, 3), np.float32)
    lib.ref_shape_hull.restype = C.c_int
    n = lib.ref_shape_hull(v.ctypes.data_as(C.POINTER(C.c_float)), len(v), t.ctypes.data_as(C.POINTER(C.c_int)), len(t),
                           out.ctypes.data_as(C.POINTER(C.c_float)), cap)
    return out[:n].copy()


def ref_world(scene, **kw):
    return World(load_ref(), "ref_", scene, **kw)


def oracle_world(scene, **kw):
    return World(load_oracle(), "ora_", scene, **kw)


def sort_contacts(c):
    """Canonical order: Bullet's own order follows DBVT traversal and is history dependent."""
    return c[np.lexsort((c["childB"], c["objB"], c["childA"], c["objA"]))]
