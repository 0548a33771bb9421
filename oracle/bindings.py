"""ctypes bindings for the two CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY, see oracle_api.h).

Importers allowed: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
  * ``load_oracle()`` -> oracle/liboracle.so   (CPU restatement; built by ``make -C oracle oracle``)
  * ``load_ref()``    -> oracle/_ref/libref.so (reference's vendored Bullet; only where it was built)
Both expose the same world API over the flat ``ora_scene`` of oracle_api.h.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

BOX, HULL, SPHERE, CYLZ = 0, 1, 2, 3
GROUP_ROBOT, GROUP_KINBODY = 1, 2
CC_NONE, CC_TIME0, CC_TIME1, CC_BETWEEN = 0, 1, 2, 3
IDENT7 = (1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


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


CONTACT_DTYPE = np.dtype([
    ("objA", "i4"), ("childA", "i4"), ("objB", "i4"), ("childB", "i4"), ("cctype", "i4"), ("time", "f4"),
    ("dist", "f8"), ("n", "f8", 3), ("ptA", "f8", 3), ("ptB", "f8", 3), ("ptB1", "f8", 3)], align=True)
assert CONTACT_DTYPE.itemsize == C.sizeof(OraContact)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Scene:
    """Flat scene: shapes, children (geometry instances) and objects (links with geometry)."""

    def __init__(self):
        self.kind, self.param, self.voff, self.vcnt = [], [], [], []
        self.verts = []
        self.child_shape, self.child_T = [], []
        self.obj_off, self.obj_cnt, self.obj_compound, self.obj_group = [], [], [], []
        self.obj_name = []

    # -- shapes ----------------------------------------------------------------------------------------
    def add_box(self, hx, hy, hz):
        return self._shape(BOX, (hx, hy, hz, 0.0), None)

    def add_sphere(self, r):
        return self._shape(SPHERE, (r, 0, 0, 0), None)

    def add_cylinder_z(self, r, half_h):
        return self._shape(CYLZ, (r, half_h, 0, 0), None)

    def add_hull(self, pts):
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        assert 1 <= len(pts) <= 64
        return self._shape(HULL, (0, 0, 0, 0), pts)

    def _shape(self, kind, param, pts):
        self.kind.append(kind)
        self.param.append(param)
        self.voff.append(sum(len(v) for v in self.verts))
        self.vcnt.append(0 if pts is None else len(pts))
        if pts is not None:
            self.verts.append(pts)
        return len(self.kind) - 1

    # -- objects ---------------------------------------------------------------------------------------
    def add_object(self, children, group, compound=None, name=None):
        """children: list of (shape_id, T7).  compound=None applies the reference's rule
        (one geometry at identity -> bare shape; bullet_collision_checker.cpp:224)."""
        if compound is None:
            compound = not (len(children) == 1 and _is_identity(children[0][1]))
        if not compound:
            assert len(children) == 1
        self.obj_off.append(len(self.child_shape))
        self.obj_cnt.append(len(children))
        self.obj_compound.append(int(compound))
        self.obj_group.append(group)
        self.obj_name.append(name or "obj%d" % len(self.obj_name))
        for s, T in children:
            self.child_shape.append(s)
            self.child_T.append(tuple(float(x) for x in T))
        return len(self.obj_off) - 1

    @property
    def n_objects(self):
        return len(self.obj_off)

    def finalize(self):
        i4, f4, f8 = np.int32, np.float32, np.float64
        a = self.arrays = dict(
            kind=np.array(self.kind, i4), param=np.array(self.param, f4).reshape(-1, 4),
            voff=np.array(self.voff, i4), vcnt=np.array(self.vcnt, i4),
            verts=(np.concatenate(self.verts).astype(f4) if self.verts else np.zeros((1, 3), f4)),
            child_shape=np.array(self.child_shape, i4), child_T=np.array(self.child_T, f8).reshape(-1, 7),
            obj_off=np.array(self.obj_off, i4), obj_cnt=np.array(self.obj_cnt, i4),
            obj_compound=np.array(self.obj_compound, i4), obj_group=np.array(self.obj_group, i4))
        for k in a:
            a[k] = np.ascontiguousarray(a[k])
        s = OraScene()
        s.n_shapes = len(a["kind"]); s.shape_kind = _p(a["kind"], C.c_int); s.shape_param = _p(a["param"], C.c_float)
        s.shape_voff = _p(a["voff"], C.c_int); s.shape_vcnt = _p(a["vcnt"], C.c_int)
        s.n_verts = len(self.verts) and int(a["verts"].shape[0]); s.verts = _p(a["verts"], C.c_float)
        s.n_children = len(a["child_shape"]); s.child_shape = _p(a["child_shape"], C.c_int)
        s.child_T = _p(a["child_T"], C.c_double)
        s.n_objects = len(a["obj_off"]); s.obj_child_off = _p(a["obj_off"], C.c_int)
        s.obj_child_cnt = _p(a["obj_cnt"], C.c_int); s.obj_compound = _p(a["obj_compound"], C.c_int)
        s.obj_group = _p(a["obj_group"], C.c_int)
        self.c = s
        return self


def _is_identity(T, e=1e-6):
    # isIdentity(), bullet_collision_checker.cpp:85-95
    return (abs(T[4]) < e and abs(T[5]) < e and abs(T[6]) < e and abs(T[0] - 1) < e and abs(T[1]) < e
            and abs(T[2]) < e and abs(T[3]) < e)


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
        self.f = f
        self.h = f("create")(C.byref(scene.c))
        self.cap = cap
        self.buf = np.zeros(cap, CONTACT_DTYPE)

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


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def load_ref():
    return C.CDLL(os.path.join(HERE, "_ref", "libref.so"))


def load_oracle():
    return C.CDLL(os.path.join(HERE, "liboracle.so"))


def ref_world(scene, **kw):
    return World(load_ref(), "ref_", scene, **kw)


def oracle_world(scene, **kw):
    return World(load_oracle(), "ora_", scene, **kw)


def sort_contacts(c):
    """Canonical order: Bullet's own order follows DBVT traversal and is history dependent."""
    return c[np.lexsort((c["childB"], c["objB"], c["childA"], c["objA"]))]
