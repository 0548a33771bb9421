"""Seeded synthetic workloads (SURVEY.md 8d / BASELINE.md 3).

The real PR2 model (OpenRAVE's robots/pr2-beta-static.zae) and the Kinect point-cloud scenes are not available
offline, so the configs use a documented synthetic "PR2-like" tree (base x,y,theta + torso + head + 2 x 7-DOF arms +
grippers, convex-decomposed links with <= 42-vertex hulls) and synthetic convex-soup obstacles: hulls of 42 support
directions of random point blobs, stored fp32 -- what btShapeHull / convex_soup.py would hand to the checker.
"""
import numpy as np

from . import model as M


def quat_from_axis_angle(axis, ang):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def T7(trans=(0, 0, 0), axis=None, ang=0.0):
    q = np.array([1.0, 0, 0, 0]) if axis is None else quat_from_axis_angle(axis, ang)
    return np.concatenate([q, np.asarray(trans, float)])


def random_T7(rng, lo, hi, rot=True):
    q = rng.normal(size=4) if rot else np.array([1.0, 0, 0, 0])
    q = q / np.linalg.norm(q)
    t = rng.uniform(lo, hi, size=3)
    return np.concatenate([q, t])


_DIRS42 = None


def dirs42():
    """42 fixed unit directions (icosahedron subdivided once), like btShapeHull's NUM_UNITSPHERE_POINTS."""
    global _DIRS42
    if _DIRS42 is None:
        t = (1 + 5 ** 0.5) / 2
        v = []
        for a in (-1, 1):
            for b in (-t, t):
                v += [(0, a, b), (a, b, 0), (b, 0, a)]
        v = np.array(v, float)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        mids = set()
        for i in range(12):
            for j in range(i + 1, 12):
                if abs(np.linalg.norm(v[i] - v[j]) - 2 / (1 + t * t) ** 0.5) < 1e-6:
                    m = v[i] + v[j]
                    mids.add(tuple(np.round(m / np.linalg.norm(m), 12)))
        _DIRS42 = np.vstack([v, np.array(sorted(mids))])
        assert _DIRS42.shape == (42, 3)
    return _DIRS42


def blob_hull(rng, scale=(0.05, 0.05, 0.15), npts=200, nverts=42, center=(0, 0, 0)):
    """Vertices (fp32) of the hull of `nverts` support directions of a random anisotropic blob."""
    if npts == 0:   # a smooth blob: its support points along the 42 directions are 42 distinct vertices (radius ~2.4 sigma +- 8 %)
        assert nverts == 42
        return (dirs42() * np.asarray(scale) * (2.4 * rng.uniform(0.92, 1.08, size=(42, 1))) + np.asarray(center)).astype(np.float32)
    pts = rng.normal(size=(npts, 3)) * np.asarray(scale) + np.asarray(center)
    if nverts == 42:
        d = dirs42()
    else:
        d = rng.normal(size=(nverts, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    idx = np.unique(np.argmax((pts - np.asarray(center)) @ d.T, axis=0))
    return pts[idx].astype(np.float32)


def random_trimesh(rng, n, kind):
    """A triangle mesh (fp32 vertices [n][3], int32 triangles [m][3]) as a GT_TriMesh geometry would hold it: kind 0 an
    anisotropic blob's surface, 1 a cloud with interior points, 2 a nearly flat slab, 3 a 24-gon prism with repeated
    points.  Triangles: the cloud's hull facets (scipy's Qhull, joggled), so interior points are named by no triangle."""
    from scipy.spatial import ConvexHull
    if kind == 0:
        p = rng.normal(size=(n, 3))
        p /= np.linalg.norm(p, axis=1)[:, None]
        p *= rng.uniform(0.05, 0.5, 3) * rng.uniform(0.8, 1.2, (n, 1))
    elif kind == 1:
        p = rng.uniform(-1, 1, (n, 3)) * rng.uniform(0.05, 0.6, 3)
    elif kind == 2:
        p = rng.uniform(-1, 1, (n, 3)) * np.array([0.5, 0.3, 1e-4])
    else:
        a = rng.integers(0, 24, n) * (2 * np.pi / 24)
        p = np.stack([0.2 * np.cos(a), 0.2 * np.sin(a), rng.choice([-0.3, 0.3], n)], 1)
    p = (p + rng.uniform(-1, 1, 3)).astype(np.float32)
    try:
        tri = ConvexHull(p.astype(np.float64), qhull_options="QJ").simplices.astype(np.int32)
    except Exception:   # degenerate for Qhull: a strip over consecutive vertices
        tri = np.stack([np.arange(n - 2), np.arange(1, n - 1), np.arange(2, n)], 1).astype(np.int32)
    return p, tri


def random_scene(rng, n_static=6, n_robot=4, compound_prob=0.5, box_prob=0.4, spread=0.6, round_prob=0.0, spheres=False):
    """A small world with mixed boxes / hulls (and, with round_prob > 0, z-cylinders; spheres only on request -- the
    engine rejects them), bare shapes and compounds.  Returns (scene, objT, robot_ids).  round_prob = 0 draws exactly the scenes of the golden fixtures."""
    s = M.Scene()
    objT, robot = [], []

    def make_children():
        ch = []
        k = 1 if rng.random() > compound_prob else int(rng.integers(1, 4))
        for i in range(k):
            if round_prob > 0 and rng.random() < round_prob:
                sh = s.add_sphere(rng.uniform(0.03, 0.2)) if (spheres and rng.random() < 0.5) else s.add_cylinder(rng.uniform(0.03, 0.15), rng.uniform(0.03, 0.25))
            elif rng.random() < box_prob:
                sh = s.add_box(*rng.uniform(0.03, 0.25, size=3))
            else:
                sh = s.add_hull(blob_hull(rng, scale=rng.uniform(0.03, 0.15, size=3), nverts=int(rng.choice([8, 20, 42]))))
            T = M.IDENT7 if (k == 1 and rng.random() < 0.6) else random_T7(rng, -0.1, 0.1)
            ch.append((sh, T))
        return ch

    for i in range(n_static):
        s.add_object(make_children(), M.GROUP_KINBODY)
        objT.append(random_T7(rng, -spread, spread))
    for i in range(n_robot):
        robot.append(s.add_object(make_children(), M.GROUP_ROBOT))
        objT.append(random_T7(rng, -spread, spread))
    s.finalize()
    return s, np.array(objT), robot


# ------------------------------------------------------------------------------------------------------------
# robots
# ------------------------------------------------------------------------------------------------------------
def boxbot(scene):
    """data/boxbot.xml: one 1 m cube link, driven by affine X|Y (collision-checker-unit.cpp:111)."""
    r = M.RobotModel()
    obj = scene.add_object([(scene.add_box(.5, .5, .5), M.IDENT7)], M.GROUP_ROBOT, name="boxlink")
    r.add_link(-1, obj=obj, name="boxlink")
    r.set_active([], M.DOF_X | M.DOF_Y)
    return r


def three_links(scene):
    """data/three_links.robot.xml: planar 3R arm about z (the base cylinder is left out)."""
    r = M.RobotModel()
    base = r.add_link(-1, name="Base")
    dims = [(0.08, 0.01, 0.005, 0.08), (0.08, 0.0075, 0.005, 0.08), (0.04, 0.005, 0.005, 0.04)]
    parent, off = base, 0.0
    for i, (hx, hy, hz, cx) in enumerate(dims):
        obj = scene.add_object([(scene.add_box(hx, hy, hz), T7((cx, 0, 0)))], M.GROUP_ROBOT, name="Arm%d" % i)
        parent = r.add_link(parent, T7((off, 0, 0)), M.J_REVOLUTE, (0, 0, 1), dof=i, obj=obj, name="Arm%d" % i)
        off = 0.16
    r.set_active([0, 1, 2])
    return r


def pr2_like(scene, seed=0, active="rightarm", npts=120):
    """Synthetic PR2-like robot: ~45 links, 19 robot DOFs.

    DOF layout: 0 torso, 1-7 right arm, 8-14 left arm, 15/16 right/left gripper, 17/18 head pan/tilt.
    active: "rightarm" | "leftarm" (K=7)  or  "fullbody" = torso + both arms + base x,y,theta (K=18; the layout
    of python_examples/fullbody_plan.py:76-80).
    """
    rng = np.random.default_rng(seed)
    r = M.RobotModel()

    def hulls(n, scale, centers, nv=42):
        ch = []
        for i in range(n):
            h = scene.add_hull(blob_hull(rng, scale=scale, nverts=nv, npts=npts if (npts or nv == 42) else 120))
            ch.append((h, T7(centers[i])))
        return ch

    def link(parent, name, T, jtype=M.J_FIXED, axis=(0, 0, 1), dof=-1, children=None, mult=1.0):
        obj = -1
        if children:
            obj = scene.add_object(children, M.GROUP_ROBOT, name=name)
        return r.add_link(parent, T, jtype, axis, dof, mult, 0.0, obj, name)

    base_children = [(scene.add_box(0.33, 0.33, 0.15), T7((0, 0, 0.15)))] + hulls(2, (0.08, 0.1, 0.1), [(-0.2, 0, 0.45), (0.2, 0, 0.35)])
    base = link(-1, "base_link", M.IDENT7, children=base_children)
    for i, (x, y) in enumerate([(0.22, 0.22), (0.22, -0.22), (-0.22, 0.22), (-0.22, -0.22)]):
        link(base, "caster%d" % i, T7((x, y, 0.05)), children=hulls(1, (0.04, 0.04, 0.03), [(0, 0, 0)], nv=20))
    link(base, "base_laser", T7((0.27, 0, 0.3)), children=[(scene.add_box(0.03, 0.03, 0.04), M.IDENT7)])
    torso = link(base, "torso_lift_link", T7((-0.05, 0, 0.74)), M.J_PRISMATIC, (0, 0, 1), 0,
                 hulls(2, (0.07, 0.12, 0.15), [(0, 0, 0.1), (0, 0, 0.4)]))
    pan = link(torso, "head_pan_link", T7((-0.02, 0, 0.38)), M.J_REVOLUTE, (0, 0, 1), 17, hulls(1, (0.05, 0.05, 0.03), [(0, 0, 0)], nv=20))
    link(pan, "head_tilt_link", T7((0.07, 0, 0)), M.J_REVOLUTE, (0, 1, 0), 18, hulls(2, (0.06, 0.1, 0.05), [(0.02, 0, 0.08), (0.05, 0, 0.15)]))
    for side, sgn, d0, gd in (("r", -1.0, 1, 15), ("l", 1.0, 8, 16)):
        p = link(torso, side + "_shoulder_pan_link", T7((0.0, sgn * 0.188, 0.0)), M.J_REVOLUTE, (0, 0, 1), d0,
                 hulls(2, (0.08, 0.08, 0.1), [(0.0, 0, -0.1), (0.05, 0, -0.25)]))
        p = link(p, side + "_shoulder_lift_link", T7((0.1, 0, 0)), M.J_REVOLUTE, (0, 1, 0), d0 + 1,
                 hulls(1, (0.06, 0.06, 0.06), [(0, 0, 0)]))
        p = link(p, side + "_upper_arm_roll_link", M.IDENT7, M.J_REVOLUTE, (1, 0, 0), d0 + 2,
                 hulls(1, (0.05, 0.04, 0.04), [(0.1, 0, 0)], nv=20))
        link(p, side + "_upper_arm_link", M.IDENT7, children=hulls(2, (0.09, 0.05, 0.05), [(0.2, 0, 0), (0.32, 0, 0)]))
        p = link(p, side + "_elbow_flex_link", T7((0.4, 0, 0)), M.J_REVOLUTE, (0, 1, 0), d0 + 3,
                 hulls(1, (0.05, 0.05, 0.05), [(0, 0, 0)]))
        p = link(p, side + "_forearm_roll_link", M.IDENT7, M.J_REVOLUTE, (1, 0, 0), d0 + 4,
                 hulls(1, (0.04, 0.04, 0.04), [(0.08, 0, 0)], nv=20))
        link(p, side + "_forearm_link", M.IDENT7, children=hulls(2, (0.07, 0.04, 0.04), [(0.16, 0, 0), (0.26, 0, 0)]))
        p = link(p, side + "_wrist_flex_link", T7((0.321, 0, 0)), M.J_REVOLUTE, (0, 1, 0), d0 + 5,
                 hulls(1, (0.03, 0.03, 0.03), [(0, 0, 0)], nv=20))
        p = link(p, side + "_wrist_roll_link", M.IDENT7, M.J_REVOLUTE, (1, 0, 0), d0 + 6)
        palm = link(p, side + "_gripper_palm_link", M.IDENT7, children=hulls(1, (0.03, 0.04, 0.02), [(0.07, 0, 0)]))
        for fs, fn in ((1.0, "l"), (-1.0, "r")):
            f = link(palm, "%s_gripper_%s_finger_link" % (side, fn), T7((0.077, fs * 0.01, 0)), M.J_REVOLUTE, (0, 0, 1), gd,
                     hulls(1, (0.025, 0.01, 0.012), [(0.04, fs * 0.012, 0)], nv=20), mult=fs)
            link(f, "%s_gripper_%s_finger_tip_link" % (side, fn), T7((0.0914, fs * 0.005, 0)), M.J_REVOLUTE, (0, 0, 1), gd,
                 hulls(1, (0.012, 0.006, 0.012), [(0.015, -fs * 0.004, 0)], nv=8), mult=-fs)
    r.n_dof = 19
    # a tucked-ish default posture for the inactive joints (arms away from the body, torso half up)
    q = np.zeros(19)
    q[0] = 0.15
    q[1:8] = [-1.8, 0.4, -1.0, -1.5, -1.1, -1.9, 3.0]
    q[8:15] = [1.8, 0.4, 1.0, -1.5, 1.1, -1.9, -3.0]
    q[15] = q[16] = 0.3
    r.q_robot = q
    if active == "rightarm":
        r.set_active(range(1, 8))
    elif active == "leftarm":
        r.set_active(range(8, 15))
    elif active == "fullbody":
        r.set_active([0] + list(range(1, 8)) + list(range(8, 15)), M.DOF_X | M.DOF_Y | M.DOF_ROTAXIS, (0, 0, 1))
    else:
        raise ValueError(active)
    return r


# ------------------------------------------------------------------------------------------------------------
# obstacles
# ------------------------------------------------------------------------------------------------------------
def add_boxes(scene, boxes, name="env"):
    """boxes: iterable of (half_extents(3), T7) -> one static KinBody link per box (OpenRAVE env XML bodies)."""
    ids = []
    for i, (he, T) in enumerate(boxes):
        ids.append(scene.add_object([(scene.add_box(*he), M.IDENT7)], M.GROUP_KINBODY, name="%s%d" % (name, i)))
    return ids, np.array([T for _, T in boxes], float).reshape(-1, 7)


def convex_soup(scene, n_hulls=4096, seed=2, extent=3.0, sigma=0.03, center=(1.5, 0.0), npts=120):
    """SURVEY config 4: ONE object with n convex children (trajoptpy/convex_soup.py makes one link with N trimesh
    geoms), each the 42-direction hull of a sigma=3 cm blob centred on a 2.5-D heightfield over extent x extent m."""
    rng = np.random.default_rng(seed)
    g = int(np.ceil(np.sqrt(n_hulls)))
    xs = (np.arange(g) + 0.5) / g * extent - extent / 2 + center[0]
    ys = (np.arange(g) + 0.5) / g * extent - extent / 2 + center[1]
    ph = rng.uniform(0, 2 * np.pi, size=4)
    ch = []
    for k in range(n_hulls):
        x, y = xs[k % g], ys[k // g]
        z = 0.75 + 0.15 * np.sin(2.1 * x + ph[0]) * np.cos(1.7 * y + ph[1]) + 0.08 * np.sin(5.3 * x + ph[2]) * np.sin(4.1 * y + ph[3])
        pts = blob_hull(rng, scale=(sigma, sigma, sigma), npts=npts, nverts=42)
        ch.append((scene.add_hull(pts), T7((x, y, z))))
    return scene.add_object(ch, M.GROUP_KINBODY, compound=True, name="soup")


def straight_line(q0, q1, T):
    return np.linspace(np.asarray(q0, float), np.asarray(q1, float), T)


# ------------------------------------------------------------------------------------------------------------
# named workloads (BASELINE.json configs)
# ------------------------------------------------------------------------------------------------------------
TABLE_BOX = ((.85, .55, .01), T7((1.1, 0, .635)))   # data/table.xml


def static_poses(scene, ids=(), poses=None):
    st = np.tile(np.array(M.IDENT7), (scene.n_objects, 1))
    if len(ids):
        st[list(ids)] = poses
    return st


def config_arm_table(seed=0):
    """Configs 1/2: PR2-like right arm (K=7) over the table of data/table.xml; d = 0.025 + 0.04."""
    s = M.Scene()
    rb = pr2_like(s, seed=seed, active="rightarm")
    ids, T = add_boxes(s, [TABLE_BOX], name="table")
    s.finalize(); rb.finalize()
    return s, rb, static_poses(s, ids, T), 0.065


def config_fullbody_soup(n_hulls=4096, seed=2, npts=120):
    """Config 4: full-body PR2-like robot (K=18: torso + 2 arms + base x,y,theta) in front of a 4096-hull convex soup
    (one object with 4096 convex children on a 3 x 3 m heightfield); d = 0.05 + 0.04.
    npts: points per random blob whose 42 support points make a hull.  With 120 (the default workload) many of the 42
    directions pick the same point: ~16 vertices per hull.  npts = 0: smooth blobs, all 42 support points distinct --
    the hull size btShapeHull gives a dense mesh (bench.py --config fullbody42)."""
    s = M.Scene()
    rb = pr2_like(s, seed=0, active="fullbody", npts=npts)
    convex_soup(s, n_hulls=n_hulls, seed=seed, center=(2.05, 0.0), npts=npts)
    s.finalize(); rb.finalize()
    return s, rb, static_poses(s), 0.09


def fullbody_trajectory(rb, rng, T=100):
    """Straight line between two seeded configurations: the base creeps along the soup's edge, arms sweep over it."""
    arm_r, arm_l = rb.q_robot[1:8], rb.q_robot[8:15]
    q0 = np.concatenate([[0.05], arm_r + rng.uniform(-0.2, 0.2, 7), arm_l + rng.uniform(-0.2, 0.2, 7),
                         [rng.uniform(-0.35, -0.15), rng.uniform(-0.8, -0.4), rng.uniform(-0.3, 0.3)]])
    reach_r = np.array([-0.3, 0.2, -0.6, -0.9, -1.0, -0.8, 1.0]) + rng.uniform(-0.3, 0.3, 7)
    reach_l = np.array([0.3, 0.2, 0.6, -0.9, 1.0, -0.8, -1.0]) + rng.uniform(-0.3, 0.3, 7)
    q1 = np.concatenate([[0.25], reach_r, reach_l, [rng.uniform(-0.05, 0.1), rng.uniform(0.4, 0.8), rng.uniform(-0.4, 0.4)]])
    return straight_line(q0, q1, T)


def add_ref_scene(scene, name):
    """One of the reference's obstacle worlds (tests/golden/ref_scenes.json, made by tests/golden/make_scenes.py from
    data/*.env.xml): one static KinBody link per box, the box geometry at its own transform inside the link -- so, like
    in the reference, every such link is a compound of one child (bullet_collision_checker.cpp:224-250)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "ref_scenes.json")) as f:
        boxes = json.load(f)[name]
    ids = []
    for b in boxes:
        T = tuple(b["quat"]) + tuple(b["translation"])
        ids.append(scene.add_object([(scene.add_box(*b["extents"]), T)], M.GROUP_KINBODY, name=b["name"]))
    return ids


def benchmark_trajectories(rb, rng, n, T=11):
    """python_examples/benchmark.py:25-30,76-77,116-118 shape: right-arm problems, straight line from a start posture to a
    joint target drawn within +-1 rad of it (the reference takes IK solutions of a grid of hand poses; IK needs
    OpenRAVE), T = 11 waypoints."""
    out = []
    for _ in range(n):
        q0 = rb.q_robot[1:8] + rng.uniform(-0.4, 0.4, size=7)
        out.append(straight_line(q0, q0 + rng.uniform(-1.0, 1.0, size=7), T))
    return np.stack(out)


def config_benchmark_scene(name="kitchen_counter"):
    """Configs 3 / 5: python_examples/benchmark.py-shaped right-arm problems (K = 7, T = 11, a discrete and a cast
    collision cost, dist_pen 0.05 => d = 0.09) in one of the reference's own box worlds (kitchen_counter: 52 boxes)."""
    s = M.Scene()
    rb = pr2_like(s, seed=0, active="rightarm")
    add_ref_scene(s, name)
    s.finalize(); rb.finalize()
    return s, rb, static_poses(s), 0.09


def config_ref_problem(name="arm_around_table"):
    """Configs 1 / 2 as the reference ships them: data/arm_around_table.json (T = 10) / arm_around_table_continuous.json
    (T = 6) -- the numbers of tests/golden/ref_problems.json, made by tests/golden/make_problems.py -- with the PR2-like right
    arm (the PR2 model itself is OpenRAVE's) over the table of data/table.xml: a static link at identity whose one box
    geometry sits at its own translation, i.e. a compound of one child (bullet_collision_checker.cpp:224-250).
    -> scene, robot, static poses, d (= dist_pen + 0.04), the init trajectory [T][7], the problem's dict."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "ref_problems.json")) as f:
        j = json.load(f)
    p, tb = j[name], j["table"]
    s = M.Scene()
    rb = pr2_like(s, seed=0, active=p["manip"])
    s.add_object([(s.add_box(*tb["extents"]), T7(tb["translation"]))], M.GROUP_KINBODY, name="table_link")
    s.finalize(); rb.finalize()
    return s, rb, static_poses(s), float(p["contact_distance"]), np.array(p["init_traj"], float), p
