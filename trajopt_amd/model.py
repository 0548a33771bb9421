"""Host-side marshalling of a collision world and a robot into flat SoA arrays.

This is what the C++ host does with OpenRAVE bodies in the reference (AddKinBody / CollisionObjectFromLink /
createShapePrimitive, bullet_collision_checker.cpp:134-255,535-561; RobotAndDOF, configuration_space.cpp),
restated over plain numpy because OpenRAVE is not available:
    shapes -> children (geometry instances) -> objects (links with geometry);  robot = flattened joint tree.
The arrays are exactly what the C-ABI upload calls take (include/trajopt_hipcc.h).
"""
import numpy as np

BOX, HULL, SPHERE, CYLZ = 0, 1, 2, 3
GROUP_ROBOT, GROUP_KINBODY = 1, 2                      # collision_checker.hpp:32-35
CC_NONE, CC_TIME0, CC_TIME1, CC_BETWEEN = 0, 1, 2, 3   # collision_checker.hpp:12-17
J_FIXED, J_REVOLUTE, J_PRISMATIC = 0, 1, 2
DOF_X, DOF_Y, DOF_Z, DOF_ROTAXIS = 1, 2, 4, 8          # OpenRAVE DOFAffine
DOF_GRABBED = -2                                       # dof of a fixed link: a grabbed body's link (TOHC_DOF_GRABBED)
IDENT7 = (1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


def _is_identity(T, e=1e-6):
    # isIdentity(), bullet_collision_checker.cpp:85-95
    return (abs(T[4]) < e and abs(T[5]) < e and abs(T[6]) < e and abs(T[0] - 1) < e and abs(T[1]) < e
            and abs(T[2]) < e and abs(T[3]) < e)


class Scene:
    """Flat scene: shapes, children (geometry instances) and objects (links with geometry)."""

    def __init__(self):
        self.kind, self.param, self.voff, self.vcnt = [], [], [], []
        self.verts = []
        self.child_shape, self.child_T = [], []
        self.obj_off, self.obj_cnt, self.obj_compound, self.obj_group = [], [], [], []
        self.obj_name = []
        self.mesh_src = {}   # shape id -> (vertices, triangles) of a GT_TriMesh geometry (add_trimesh)
        self.arrays = None

    # -- shapes ----------------------------------------------------------------------------------------
    def add_box(self, hx, hy, hz):
        """Half extents, as OpenRAVE's GetBoxExtents() (bullet_collision_checker.cpp:148)."""
        return self._shape(BOX, (hx, hy, hz, 0.0), None)

    def add_sphere(self, radius):
        """GT_Sphere -> btSphereShape(radius) (bullet_collision_checker.cpp:150-152)."""
        return self._shape(SPHERE, (radius, 0.0, 0.0, 0.0), None)

    def add_cylinder(self, radius, half_height):
        """GT_Cylinder -> btCylinderShapeZ(r, r, height / 2) (:153-158): param = (radius, HALF height)."""
        return self._shape(CYLZ, (radius, half_height, 0.0, 0.0), None)

    def add_hull(self, pts):
        """Convex hull of <= 64 points, exactly the points addPoint() would receive (:191-202)."""
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        if not 1 <= len(pts) <= 64:
            raise ValueError("hull shapes hold 1..64 vertices (the reference reduces meshes to <= 42)")
        return self._shape(HULL, (0, 0, 0, 0), pts)

    def add_trimesh(self, vertices, triangles):
        """GT_TriMesh (bullet_collision_checker.cpp:160-203): below 50 vertices the mesh's vertices are the hull's points,
        from 50 on btShapeHull's reduction to <= 42 points (engine.shape_hull = tohc_shape_hull)."""
        from . import engine
        sid = self.add_hull(engine.shape_hull(vertices, triangles))
        self.mesh_src[sid] = (np.ascontiguousarray(vertices, np.float32).reshape(-1, 3), np.ascontiguousarray(triangles, np.int32).reshape(-1, 3))
        return sid

    def _shape(self, kind, param, pts):
        self.kind.append(kind)
        self.param.append(tuple(float(x) for x in param))
        self.voff.append(sum(len(v) for v in self.verts))
        self.vcnt.append(0 if pts is None else len(pts))
        if pts is not None:
            self.verts.append(pts)
        return len(self.kind) - 1

    # -- objects ---------------------------------------------------------------------------------------
    def add_object(self, children, group, compound=None, name=None):
        """children: list of (shape_id, T7).  compound=None applies the reference's rule
        (one geometry at identity -> bare shape, else btCompoundShape; bullet_collision_checker.cpp:224-250)."""
        if compound is None:
            compound = not (len(children) == 1 and _is_identity(children[0][1]))
        if not compound and len(children) != 1:
            raise ValueError("a bare convex object has exactly one child")
        self.obj_off.append(len(self.child_shape))
        self.obj_cnt.append(len(children))
        self.obj_compound.append(int(compound))
        self.obj_group.append(int(group))
        self.obj_name.append(name or "obj%d" % len(self.obj_name))
        for s, T in children:
            self.child_shape.append(int(s))
            self.child_T.append(tuple(float(x) for x in T))
        return len(self.obj_off) - 1

    @property
    def n_objects(self):
        return len(self.obj_off)

    def finalize(self):
        i4, f4, f8 = np.int32, np.float32, np.float64
        a = dict(
            kind=np.array(self.kind, i4), param=np.array(self.param, f4).reshape(-1, 4),
            voff=np.array(self.voff, i4), vcnt=np.array(self.vcnt, i4),
            verts=(np.concatenate(self.verts).astype(f4) if self.verts else np.zeros((1, 3), f4)),
            child_shape=np.array(self.child_shape, i4), child_T=np.array(self.child_T, f8).reshape(-1, 7),
            obj_off=np.array(self.obj_off, i4), obj_cnt=np.array(self.obj_cnt, i4),
            obj_compound=np.array(self.obj_compound, i4), obj_group=np.array(self.obj_group, i4))
        self.arrays = {k: np.ascontiguousarray(v) for k, v in a.items()}
        return self


def scene_from_arrays(arrays):
    """Rebuild a finalized Scene from its flat arrays (e.g. a tests/golden fixture)."""
    s = Scene()
    s.arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    a = s.arrays
    s.obj_off, s.obj_cnt = a["obj_off"].tolist(), a["obj_cnt"].tolist()
    s.obj_compound, s.obj_group = a["obj_compound"].tolist(), a["obj_group"].tolist()
    s.child_shape = a["child_shape"].tolist()
    s.kind = a["kind"].tolist()
    s.obj_name = ["obj%d" % i for i in range(len(s.obj_off))]
    return s


class RobotModel:
    """Flattened kinematic tree + the active DOFs of a RobotAndDOF (configuration_space.hpp).

    Conventions (OpenRAVE itself is absent; see oracle/oracle_api.h): link i hangs off parent[i] (parents first),
    T_world_link = T_world_parent * T_origin[i] * Motion(jtype, axis, mult*q[dof]+offs); the optimisation vector is
    [q[joint_inds]..., affine X, Y, Z, RotationAxis].
    """

    def __init__(self):
        self.parent, self.jtype, self.T_origin, self.axis = [], [], [], []
        self.dof, self.mult, self.offs, self.link_obj, self.link_name = [], [], [], [], []
        self.n_dof = 0
        self.q_robot = None
        self.base_T = IDENT7
        self.joint_inds = []
        self.affine = 0
        self.rot_axis = (0.0, 0.0, 1.0)
        self.arrays = None

    def add_link(self, parent, T_origin=IDENT7, jtype=J_FIXED, axis=(0, 0, 1), dof=-1, mult=1.0, offs=0.0, obj=-1,
                 name=None):
        if parent >= len(self.parent):
            raise ValueError("parents must precede children")
        ax = np.asarray(axis, float)
        ax = ax / np.linalg.norm(ax)
        self.parent.append(int(parent)); self.jtype.append(int(jtype))
        self.T_origin.append(tuple(float(x) for x in T_origin)); self.axis.append(tuple(ax))
        self.dof.append(int(dof)); self.mult.append(float(mult)); self.offs.append(float(offs))
        self.link_obj.append(int(obj)); self.link_name.append(name or "link%d" % len(self.link_name))
        if dof >= 0:
            self.n_dof = max(self.n_dof, dof + 1)
        return len(self.parent) - 1

    def grab(self, grabber, T_in_grabber, obj, name=None):
        """robot->Grab(body, grabberLink): one link of the grabbed body (its object `obj`), rigidly attached to link
        `grabber` at pose T_in_grabber.  GetAffectedLinks lists it whether or not an active DOF moves it, and its Jacobian
        is the grabber's (configuration_space.cpp:120-131)."""
        return self.add_link(grabber, T_in_grabber, J_FIXED, dof=DOF_GRABBED, obj=obj, name=name or "grabbed%d" % len(self.parent))

    def set_active(self, joint_inds, affine=0, rot_axis=(0, 0, 1)):
        """RobotAndDOF(robot, joint_inds, affinedofs, rotationaxis)."""
        self.joint_inds = [int(j) for j in joint_inds]
        self.affine = int(affine)
        self.rot_axis = tuple(float(x) for x in rot_axis)
        return self

    @property
    def n_links(self):
        return len(self.parent)

    @property
    def K(self):
        return len(self.joint_inds) + bin(self.affine & 15).count("1")

    def finalize(self):
        i4, f8 = np.int32, np.float64
        q = np.zeros(max(self.n_dof, 1), f8) if self.q_robot is None else np.asarray(self.q_robot, f8)
        a = dict(
            parent=np.array(self.parent, i4), jtype=np.array(self.jtype, i4),
            T_origin=np.array(self.T_origin, f8).reshape(-1, 7), axis=np.array(self.axis, f8).reshape(-1, 3),
            dof=np.array(self.dof, i4), mult=np.array(self.mult, f8), offs=np.array(self.offs, f8),
            link_obj=np.array(self.link_obj, i4), q_robot=q, base_T=np.array(self.base_T, f8),
            joint_inds=np.array(self.joint_inds if self.joint_inds else [0], i4)[:len(self.joint_inds) or 1],
            rot_axis=np.array(self.rot_axis, f8))
        self.arrays = {k: np.ascontiguousarray(v) for k, v in a.items()}
        return self

    def affected_links(self):
        """GetAffectedLinks(only_with_geom=True): (object ids, link indices) in link order
        (configuration_space.cpp:110-118)."""
        active = set(self.joint_inds)
        objs, links = [], []
        for i in range(self.n_links):
            if self.link_obj[i] < 0:
                continue
            aff = self.affine > 0 or self.dof[i] == DOF_GRABBED   # grabbed links are listed unconditionally (:120-131)
            j = i
            while j >= 0 and not aff:
                if self.jtype[j] != J_FIXED and self.dof[j] in active:
                    aff = True
                j = self.parent[j]
            if aff:
                objs.append(self.link_obj[i]); links.append(i)
        return objs, links
