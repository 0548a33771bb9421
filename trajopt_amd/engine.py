"""ctypes binding of libtrajopt_hipcc.so (include/trajopt_hipcc.h) -- the Python-side test and bench harness.

Everything goes through the C-ABI; there is no CPU fallback: if the library or a GPU is missing, construction
raises.  The host-side mirror of trajopt::CollisionChecker built on this lives in collision_checker.py.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOHC_LIB") or os.path.join(HERE, "libtrajopt_hipcc.so")   # TOHC_LIB: experiment builds (scripts/)

RECORD_DTYPE = np.dtype([
    ("problem", "i4"), ("t", "i4"), ("idA", "i4"), ("childA", "i4"), ("idB", "i4"), ("childB", "i4"),
    ("cctype", "i4"), ("time", "f4"), ("dist", "f8"), ("n", "f8", 3), ("ptA", "f8", 3), ("ptB", "f8", 3),
    ("ptB1", "f8", 3), ("const0", "f8"), ("const1", "f8")], align=True)
assert RECORD_DTYPE.itemsize == 152


class Counters(C.Structure):
    _fields_ = [("aabb_tests", C.c_int64), ("pair_evals", C.c_int64), ("epa_evals", C.c_int64),
                ("contacts", C.c_int64), ("ms_last_call", C.c_double),
                ("cum_aabb_tests", C.c_int64), ("cum_pair_evals", C.c_int64), ("cum_epa_evals", C.c_int64),
                ("cum_launches", C.c_int64), ("cum_ms_fk", C.c_double), ("cum_ms_children", C.c_double),
                ("cum_ms_broadphase", C.c_double), ("cum_ms_gjk", C.c_double), ("cum_ms_epa", C.c_double),
                ("gjk_evals", C.c_int64), ("cum_gjk_evals", C.c_int64), ("cum_ms_finish", C.c_double),
                ("gjk_iters", C.c_int64), ("pen_supports", C.c_int64), ("epa_supports", C.c_int64),
                ("cum_gjk_iters", C.c_int64), ("cum_pen_supports", C.c_int64), ("cum_epa_supports", C.c_int64),
                ("epa_redone", C.c_int64), ("cum_epa_redone", C.c_int64),
                ("pen_refiled", C.c_int64), ("cum_pen_refiled", C.c_int64)]


class Hinges(C.Structure):   # tohc_hinges
    _fields_ = [("n_problems", C.c_int32), ("T", C.c_int32), ("K", C.c_int32), ("gap", C.c_int32), ("n_rows", C.c_int64), ("nnz", C.c_int64),
                ("row_off", C.POINTER(C.c_int64)), ("rhs", C.POINTER(C.c_double)), ("dist", C.POINTER(C.c_double)), ("row_t", C.POINTER(C.c_int32)),
                ("col_off", C.POINTER(C.c_int64)), ("col_idx", C.POINTER(C.c_int32)), ("col_val", C.POINTER(C.c_double))]


HINGE_CAST = 1 << 30


class EngineError(RuntimeError):
    """The reference throws std::runtime_error (collision_checker.hpp:57-62); the C-ABI returns codes."""

    def __init__(self, code, msg):
        super().__init__("tohc error %d: %s" % (code, msg))
        self.code = code


_lib = None

_i32p, _f32p, _f64p, _u8p = (C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8))

SYMBOLS = {
    "tohc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "tohc_destroy": (None, [C.c_void_p]),
    "tohc_last_error": (C.c_char_p, [C.c_void_p]),
    "tohc_upload_scene": (C.c_int, [C.c_void_p, C.c_int, _i32p, _f32p, _i32p, _i32p, C.c_int, _f32p, _f32p, _f32p, C.c_int,
                                    _i32p, _f64p, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p]),
    "tohc_upload_robot": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, _f64p, _f64p, _i32p, _f64p, _f64p, _i32p, C.c_int,
                                    _f64p, _f64p, C.c_int, _i32p, C.c_int, _f64p]),
    "tohc_set_allowed_matrix": (C.c_int, [C.c_void_p, C.c_int, _u8p]),
    "tohc_set_contact_distance": (C.c_int, [C.c_void_p, C.c_float]),
    "tohc_get_contact_distance": (C.c_double, [C.c_void_p]),
    "tohc_links_vs_all_batch": (C.c_int, [C.c_void_p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                          C.POINTER(_f64p), C.POINTER(C.c_int)]),
    "tohc_cast_vs_all_batch": (C.c_int, [C.c_void_p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(_f64p), C.POINTER(C.c_int)]),
    "tohc_stage_dofs": (C.c_int, [C.c_void_p, _f64p, C.c_int, C.c_int]),
    "tohc_enqueue_links_vs_all": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tohc_enqueue_cast_vs_all": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tohc_enqueue_linearization": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tohc_enqueue_linearization_window": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tohc_sync": (C.c_int, [C.c_void_p]),
    "tohc_fetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(_f64p), C.POINTER(C.c_int)]),
    "tohc_fetch_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tohc_fetch_hinges": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.POINTER(Hinges)]),
    "tohc_fetch_hinges_begin": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "tohc_fetch_hinges_end": (C.c_int, [C.c_void_p, C.POINTER(Hinges)]),
    "tohc_links_vs_all_poses": (C.c_int, [C.c_void_p, _f64p, C.c_int, _i32p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_int)]),
    "tohc_cast_vs_all_poses": (C.c_int, [C.c_void_p, _f64p, _f64p, C.c_int, _i32p, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_int)]),
    "tohc_all_vs_all_poses": (C.c_int, [C.c_void_p, _f64p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "tohc_continuous_check_trajectory": (C.c_int, [C.c_void_p, _f64p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "tohc_continuous_check_poses": (C.c_int, [C.c_void_p, _f64p, C.c_int, _i32p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "tohc_forward_kinematics": (C.c_int, [C.c_void_p, _f64p, C.c_int, _f64p]),
    "tohc_ray_cast": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, C.c_int, C.POINTER(C.c_float)]),
    "tohc_ray_cast_poses": (C.c_int, [C.c_void_p, _f64p, _f64p, _f64p, C.c_int, C.POINTER(C.c_float)]),
    "tohc_ray_cast_poses_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),   # device pointers
    "tohc_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "tohc_set_capacity": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "tohc_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tohc_shape_hull": (C.c_int, [_f32p, C.c_int, _i32p, C.c_int, _f32p, C.c_int]),
}


def load_library():
    """Load the C-ABI library and declare every symbol of include/trajopt_hipcc.h (raises if one is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(-2, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def shape_hull(vertices, triangles):
    """The points createShapePrimitive gives a GT_TriMesh geometry's btConvexHullShape (bullet_collision_checker.cpp:
    174-203): the mesh's own vertices below 50 of them, else btShapeHull's <= 42 points (tohc_shape_hull; host code, no
    GPU needed)."""
    lib = load_library()
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
    n = lib.tohc_shape_hull(_p(v, C.c_float), len(v), _p(t, C.c_int32), len(t), None, 0)
    if n < 0:
        raise EngineError(n, "tohc_shape_hull: bad mesh (no vertices, no triangles, or an index outside the vertex table)")
    out = np.zeros((n, 3), np.float32)
    lib.tohc_shape_hull(_p(v, C.c_float), len(v), _p(t, C.c_int32), len(t), _p(out, C.c_float), n)
    return out


class Engine:
    def __init__(self, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.tohc_create(C.byref(self.h), int(device))
        if rc != 0:
            self.h = None
            raise EngineError(rc, "tohc_create failed (no HIP device? the engine has no CPU fallback)")
        self.K = 0
        self.n_objects = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tohc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise EngineError(rc, self.lib.tohc_last_error(self.h).decode())

    # -- uploads ---------------------------------------------------------------------------------------
    def upload_scene(self, scene, static_T):
        a = scene.arrays
        static_T = np.ascontiguousarray(static_T, np.float64).reshape(-1, 7)
        assert len(static_T) == len(a["obj_off"])
        v = np.ascontiguousarray(a["verts"], np.float32).reshape(-1, 3)
        nv = int(a["vcnt"].sum())
        vx, vy, vz = (np.ascontiguousarray(v[:, i]) for i in range(3))
        self._chk(self.lib.tohc_upload_scene(
            self.h, len(a["kind"]), _p(a["kind"], C.c_int32), _p(a["param"], C.c_float), _p(a["voff"], C.c_int32),
            _p(a["vcnt"], C.c_int32), nv, _p(vx, C.c_float), _p(vy, C.c_float), _p(vz, C.c_float),
            len(a["child_shape"]), _p(a["child_shape"], C.c_int32), _p(a["child_T"], C.c_double), len(a["obj_off"]),
            _p(a["obj_off"], C.c_int32), _p(a["obj_cnt"], C.c_int32), _p(a["obj_compound"], C.c_int32),
            _p(a["obj_group"], C.c_int32), _p(static_T, C.c_double)))
        self.n_objects = len(a["obj_off"])

    def upload_robot(self, robot):
        a = robot.arrays
        self._chk(self.lib.tohc_upload_robot(
            self.h, robot.n_links, _p(a["parent"], C.c_int32), _p(a["jtype"], C.c_int32), _p(a["T_origin"], C.c_double),
            _p(a["axis"], C.c_double), _p(a["dof"], C.c_int32), _p(a["mult"], C.c_double), _p(a["offs"], C.c_double),
            _p(a["link_obj"], C.c_int32), robot.n_dof, _p(a["q_robot"], C.c_double), _p(a["base_T"], C.c_double),
            len(robot.joint_inds), _p(a["joint_inds"], C.c_int32), robot.affine, _p(a["rot_axis"], C.c_double)))
        self.K = robot.K
        self.n_links = robot.n_links

    def set_allowed_matrix(self, m):
        m = np.ascontiguousarray(m, np.uint8)
        self._chk(self.lib.tohc_set_allowed_matrix(self.h, m.shape[0], _p(m, C.c_uint8)))

    def set_contact_distance(self, d):
        self._chk(self.lib.tohc_set_contact_distance(self.h, float(d)))

    def get_contact_distance(self):
        return self.lib.tohc_get_contact_distance(self.h)

    def set_kernel_timing(self, on):
        """Per-stage HIP events between the kernels on / off (tohc_set_kernel_timing): off for the lowest latency."""
        self._chk(self.lib.tohc_set_kernel_timing(self.h, int(bool(on))))

    def set_capacity(self, max_pairs, max_records):
        self._chk(self.lib.tohc_set_capacity(self.h, int(max_pairs), int(max_records)))

    # -- queries ---------------------------------------------------------------------------------------
    def _collect(self, rec_p, grad_p, n, K, copy):
        n = n.value
        if n == 0:
            return np.zeros(0, RECORD_DTYPE), np.zeros((0, 2, max(K, 1)))
        recs = np.frombuffer((C.c_char * (n * 152)).from_address(rec_p.value), dtype=RECORD_DTYPE, count=n)
        grads = None
        if grad_p is not None and bool(grad_p):
            grads = np.ctypeslib.as_array(grad_p, shape=(n, 2, K))
        if copy:
            recs = recs.copy()
            grads = None if grads is None else grads.copy()
        return recs, grads

    def links_vs_all_batch(self, dofs, B, T, filter_mask=-1, want_jacobians=True, copy=True):
        dofs = np.ascontiguousarray(dofs, np.float64)
        assert dofs.size == B * T * self.K
        rec, g, n = C.c_void_p(), _f64p(), C.c_int()
        self._chk(self.lib.tohc_links_vs_all_batch(self.h, _p(dofs, C.c_double), B, T, int(filter_mask), int(want_jacobians),
                                                   C.byref(rec), C.byref(g), C.byref(n)))
        return self._collect(rec, g, n, self.K, copy)

    def cast_vs_all_batch(self, dofs, B, T, gap=1, want_jacobians=True, copy=True):
        dofs = np.ascontiguousarray(dofs, np.float64)
        assert dofs.size == B * T * self.K
        rec, g, n = C.c_void_p(), _f64p(), C.c_int()
        self._chk(self.lib.tohc_cast_vs_all_batch(self.h, _p(dofs, C.c_double), B, T, int(gap), int(want_jacobians),
                                                  C.byref(rec), C.byref(g), C.byref(n)))
        return self._collect(rec, g, n, self.K, copy)

    # resident pieces: stage once, enqueue many times, fetch when the host needs the slab
    def stage_dofs(self, dofs, B, T):
        dofs = np.ascontiguousarray(dofs, np.float64)
        assert dofs.size == B * T * self.K
        self._chk(self.lib.tohc_stage_dofs(self.h, _p(dofs, C.c_double), B, T))

    def enqueue_links_vs_all(self, filter_mask=-1, want_jacobians=True, append=False):
        self._chk(self.lib.tohc_enqueue_links_vs_all(self.h, int(filter_mask), int(want_jacobians), int(append)))

    def enqueue_cast_vs_all(self, gap=1, want_jacobians=True, append=False):
        self._chk(self.lib.tohc_enqueue_cast_vs_all(self.h, int(gap), int(want_jacobians), int(append)))

    def enqueue_linearization(self, filter_mask=-1, gap=1, want_jacobians=True):
        """A new slab = enqueue_links_vs_all(filter_mask, append=False) + enqueue_cast_vs_all(gap, append=True), with one
        broadphase launch for both passes where the engine can (tohc_enqueue_linearization)."""
        self._chk(self.lib.tohc_enqueue_linearization(self.h, int(filter_mask), int(gap), int(want_jacobians)))

    def enqueue_linearization_window(self, n_discrete, filter_mask=-1, gap=1, want_jacobians=True):
        """One timestep shard's linearization: the discrete pass over the first n_discrete staged timesteps (the staged count
        T, or T - gap when the last `gap` poses are the halo), the cast pass over every staged pair
        (tohc_enqueue_linearization_window; multi_gpu.timestep_shard says what to stage)."""
        self._chk(self.lib.tohc_enqueue_linearization_window(self.h, int(filter_mask), int(gap), int(want_jacobians), int(n_discrete)))

    def sync(self):
        self._chk(self.lib.tohc_sync(self.h))

    def fetch(self, copy=True):
        rec, g, n = C.c_void_p(), _f64p(), C.c_int()
        self._chk(self.lib.tohc_fetch(self.h, C.byref(rec), C.byref(g), C.byref(n)))
        return self._collect(rec, g, n, self.K, copy)

    @staticmethod
    def _hinges(h, copy):
        """tohc_hinges -> dict of numpy arrays (views of the engine's pinned block unless copy)"""
        ncols = h.n_problems * h.T * h.K
        def arr(p, n, dt):
            a = np.ctypeslib.as_array(p, shape=(n,)) if n > 0 else np.zeros(0, dt)
            return a.copy() if copy else a
        return dict(B=h.n_problems, T=h.T, K=h.K, gap=h.gap, n_rows=int(h.n_rows), nnz=int(h.nnz),
                    row_off=arr(h.row_off, h.n_problems + 1, np.int64), rhs=arr(h.rhs, h.n_rows, np.float64), dist=arr(h.dist, h.n_rows, np.float64),
                    row_t=arr(h.row_t, h.n_rows, np.int32), col_off=arr(h.col_off, ncols + 1, np.int64), col_idx=arr(h.col_idx, h.nnz, np.int32),
                    col_val=arr(h.col_val, h.nnz, np.float64))

    def fetch_hinges(self, dist_pen_discrete, dist_pen_cast=None, copy=True):
        """The collision hinges of the enqueued linearization in BPMPD's column layout (include/trajopt_hipcc.h: tohc_hinges)."""
        h = Hinges()
        self._chk(self.lib.tohc_fetch_hinges(self.h, float(dist_pen_discrete), float(dist_pen_discrete if dist_pen_cast is None else dist_pen_cast), C.byref(h)))
        return self._hinges(h, copy)

    def fetch_hinges_begin(self, dist_pen_discrete, dist_pen_cast=None):
        self._chk(self.lib.tohc_fetch_hinges_begin(self.h, float(dist_pen_discrete), float(dist_pen_discrete if dist_pen_cast is None else dist_pen_cast)))

    def fetch_hinges_end(self, copy=True):
        h = Hinges()
        self._chk(self.lib.tohc_fetch_hinges_end(self.h, C.byref(h)))
        return self._hinges(h, copy)

    def fetch_device(self):
        """-> (device address of the records, device address of the gradient rows or 0, n, doubles per gradient row)"""
        rec, g, n, row = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        self._chk(self.lib.tohc_fetch_device(self.h, C.byref(rec), C.byref(g), C.byref(n), C.byref(row)))
        return rec.value or 0, g.value or 0, n.value, row.value

    def links_vs_all_poses(self, obj_T, query, filter_mask=-1):
        obj_T = np.ascontiguousarray(obj_T, np.float64).reshape(-1, self.n_objects, 7)
        q = np.ascontiguousarray(query, np.int32)
        rec, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.tohc_links_vs_all_poses(self.h, _p(obj_T, C.c_double), obj_T.shape[0], _p(q, C.c_int32), len(q),
                                                   int(filter_mask), C.byref(rec), C.byref(n)))
        return self._collect(rec, None, n, 1, True)[0]

    def all_vs_all_poses(self, obj_T):
        """AllVsAll (bullet_collision_checker.cpp:463-498) at the given object poses: every unordered object pair once, contacts
        up to 2.001 d, values after the manifold round trip; every record's weight is 1 (tohc_all_vs_all_poses)."""
        obj_T = np.ascontiguousarray(obj_T, np.float64).reshape(self.n_objects, 7)
        rec, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.tohc_all_vs_all_poses(self.h, _p(obj_T, C.c_double), C.byref(rec), C.byref(n)))
        return self._collect(rec, None, n, 1, True)[0]

    def cast_vs_all_poses(self, obj_T, query, T1):
        obj_T = np.ascontiguousarray(obj_T, np.float64).reshape(-1, self.n_objects, 7)
        q = np.ascontiguousarray(query, np.int32)
        T1 = np.ascontiguousarray(T1, np.float64).reshape(obj_T.shape[0], len(q), 7)
        rec, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.tohc_cast_vs_all_poses(self.h, _p(obj_T, C.c_double), _p(T1, C.c_double), obj_T.shape[0],
                                                  _p(q, C.c_int32), len(q), C.byref(rec), C.byref(n)))
        return self._collect(rec, None, n, 1, True)[0]

    def continuous_check_trajectory(self, dofs, B, T):
        """ContinuousCheckTrajectory for B trajectories of T waypoints -> records (time = hit fraction of segment t)"""
        dofs = np.ascontiguousarray(dofs, np.float64)
        assert dofs.size == B * T * self.K
        rec, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.tohc_continuous_check_trajectory(self.h, _p(dofs, C.c_double), B, T, C.byref(rec), C.byref(n)))
        return self._collect(rec, None, n, 1, True)[0]

    def continuous_check_poses(self, obj_T, query):
        obj_T = np.ascontiguousarray(obj_T, np.float64).reshape(-1, self.n_objects, 7)
        q = np.ascontiguousarray(query, np.int32)
        rec, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.tohc_continuous_check_poses(self.h, _p(obj_T, C.c_double), obj_T.shape[0], _p(q, C.c_int32), len(q),
                                                       C.byref(rec), C.byref(n)))
        return self._collect(rec, None, n, 1, True)[0]

    def ray_cast(self, dofs, p_from, p_to):
        """RayCastCollision for a batch of rays at the robot configuration `dofs` -> float32[n] closest hit fractions
        (1 = nothing hit; the reference's bool is `< 1`)."""
        return self._rays(self.lib.tohc_ray_cast, np.ascontiguousarray(dofs, np.float64).reshape(self.K), p_from, p_to)

    def ray_cast_poses(self, obj_T, p_from, p_to):
        return self._rays(self.lib.tohc_ray_cast_poses, np.ascontiguousarray(obj_T, np.float64).reshape(self.n_objects, 7), p_from, p_to)

    def ray_cast_poses_noalloc(self, d_obj_T, d_from, d_to, n_rays, d_out):
        """The allocation-free variant: every argument is a DEVICE address (int) of caller-owned memory."""
        self._chk(self.lib.tohc_ray_cast_poses_noalloc(self.h, C.c_void_p(d_obj_T), C.c_void_p(d_from), C.c_void_p(d_to), int(n_rays), C.c_void_p(d_out)))

    def _rays(self, fn, state, p_from, p_to):
        a = np.ascontiguousarray(p_from, np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(p_to, np.float64).reshape(-1, 3)
        if len(a) != len(b):
            raise ValueError("ray_cast: %d start points, %d end points" % (len(a), len(b)))
        out = np.ones(len(a), np.float32)
        self._chk(fn(self.h, _p(state, C.c_double), _p(a, C.c_double), _p(b, C.c_double), len(a), _p(out, C.c_float)))
        return out

    def forward_kinematics(self, dofs):
        dofs = np.ascontiguousarray(dofs, np.float64).reshape(-1, self.K)
        out = np.zeros((len(dofs), self.n_links, 7))
        self._chk(self.lib.tohc_forward_kinematics(self.h, _p(dofs, C.c_double), len(dofs), _p(out, C.c_double)))
        return out

    def counters(self):
        c = Counters()
        self._chk(self.lib.tohc_get_counters(self.h, C.byref(c)))
        return {name: getattr(c, name) for name, _ in Counters._fields_}
