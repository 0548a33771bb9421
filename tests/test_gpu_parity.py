"""GPU suite (-m gpu): the HIP engine, called through the C-ABI, against the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star): contact pair set bit-exact; distances / normals / points compared BIT-FOR-BIT with the
oracle (stricter than the 1e-5 m / 1e-4 rad asked); Jacobian rows within 1e-5 relative (FK runs on the device with
the device's sin/cos).  Full-size configs are checked through size-independent properties.
"""
import glob
import os

import numpy as np
import pytest

from oracle import bindings as B
from trajopt_amd import model as M, synth
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "random_scene_*.npz")))


@pytest.fixture(scope="module")
def eng(built):
    from trajopt_amd import engine
    assert os.path.exists(engine.LIB_PATH), "HIP extension missing: the product path must not fall back"
    e = engine.Engine(0)
    yield e
    e.close()


def T(x, y=0, z=0):
    return (1, 0, 0, 0, x, y, z)


def test_reference_box_distance_counts(eng):
    # collision-checker-unit.cpp:48-87 through the C-ABI (BodyVsAll(box0))
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    o0 = s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    s.finalize()
    eng.upload_scene(s, [T(0), T(0)])
    for d, x, n in [(0, .9, 1), (0, 1.1, 0), (.04, 1.1, 0), (.1, 1.09, 1), (.2, 1.1, 1)]:
        eng.set_contact_distance(d)
        assert eng.get_contact_distance() == np.float32(d)
        rec = eng.links_vs_all_poses([T(0), T(x)], [o0])
        assert len(rec) == n, (d, x)
    eng.set_contact_distance(0)
    rec = eng.links_vs_all_poses([T(0), T(.9)], [o0])
    assert np.float32(rec[0]["dist"]) == np.float32(-0.0577350222)  # SURVEY.md 8c known answer (EPA, diagonal normal)


def test_reference_cast_known_answers(eng):
    # collision-checker-unit.cpp:130-150
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    bot = s.add_object([(b, M.IDENT7)], M.GROUP_ROBOT)
    s.finalize()
    eng.upload_scene(s, [T(0), T(0)])
    eng.set_contact_distance(0.05)
    cases = [((-1.9, 0), (0, 1.9), (-1 / np.sqrt(2), 1 / np.sqrt(2), 0), .5, 1e-1),
             ((0, .9), (0, 2), (0, 1, 0), 0, 1e-6), ((0, 2), (0, .9), (0, 1, 0), 1, 1e-6)]
    for a, b_, nrm, time, ttol in cases:
        rec = eng.cast_vs_all_poses([T(0), T(*a)], [bot], [T(*b_)])
        assert len(rec) == 1
        robot_normal = (1 if rec[0]["idA"] == bot else -1) * rec[0]["n"]
        assert np.allclose(robot_normal, nrm, atol=1e-4) and abs(rec[0]["time"] - time) <= ttol


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures_bit_exact(eng, path):
    z = np.load(path, allow_pickle=False)
    s = M.scene_from_arrays({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
    eng.upload_scene(s, z["objT"])
    robot = z["robot"]
    for i, d in enumerate(z["dists"]):
        eng.set_contact_distance(float(d))
        rec = eng.links_vs_all_poses(z["objT"], robot)
        util.assert_same_contacts(util.records_to_contacts(rec), z["disc_%d" % i], "disc d=%g" % d)
        rec = eng.cast_vs_all_poses(z["objT"], robot, z["T1"])
        util.assert_same_contacts(util.records_to_contacts(rec), z["cast_%d" % i], "cast d=%g" % d)


@pytest.mark.parametrize("seed,spread", [(200, 0.6), (201, 0.3), (202, 0.25), (203, 0.45), (204, 0.3), (205, 1.0), (320, 0.3), (321, 0.5), (322, 0.25),
                                         (940, 0.35), (941, 0.3), (942, 0.5), (943, 0.25)])
def test_random_scenes_vs_oracle_poses_api(eng, seed, spread):
    """Several pose sets per call (the batch axis), discrete and cast, dense (EPA heavy) and sparse.
    Seeds >= 300: half of the shapes are z-cylinders (GT_Cylinder, bullet_collision_checker.cpp:153-158).
    Seeds >= 900: ~45 % spheres (GT_Sphere, :150-152) -- GJK / EPA with the radius as margin, the analytic sphere-sphere
    algorithm, swept spheres, the unexpanded child-box cull of a swept sphere against a compound's sphere child."""
    rng = np.random.default_rng(seed)
    if seed >= 900:
        s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=6, round_prob=0.9, spheres=True)
    else:
        s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=5, round_prob=0.5 if seed >= 300 else 0.0)
    nsets = 4
    sets = np.stack([objT] + [np.array([synth.random_T7(rng, -spread, spread) for _ in range(len(objT))]) for _ in range(nsets - 1)])
    T1 = sets[:, robot].copy()
    T1[..., 4:] += rng.normal(size=T1[..., 4:].shape) * 0.15
    T1[..., :4] += rng.normal(size=T1[..., :4].shape) * 0.2
    T1[..., :4] /= np.linalg.norm(T1[..., :4], axis=-1, keepdims=True)
    eng.upload_scene(s, objT)
    w = B.oracle_world(s)
    for d in (0.0, 0.09):
        eng.set_contact_distance(d)
        w.set_contact_distance(d)
        rec = eng.links_vs_all_poses(sets, robot)
        pe_d = eng.counters()["pair_evals"]
        recc = eng.cast_vs_all_poses(sets, robot, T1)
        pe_c = eng.counters()["pair_evals"]
        c0 = w.counter(0)
        for k in range(nsets):
            util.assert_same_contacts(util.records_to_contacts(rec[rec["problem"] == k]), w.links_vs_all(sets[k], robot), "disc set %d" % k)
        c1 = w.counter(0)
        for k in range(nsets):
            util.assert_same_contacts(util.records_to_contacts(recc[recc["problem"] == k]),
                                      w.cast_vs_all(sets[k], robot, sets[k][robot], T1[k]), "cast set %d" % k)
        # the unit of the headline metric: both sides count the same pair evaluations (box-cull survivors)
        assert (pe_d, pe_c) == (c1 - c0, w.counter(0) - c1)
        # AllVsAll (bullet_collision_checker.cpp:463-498): every unordered object pair once, contacts up to 2.001 d, values after the
        # manifold round trip -- bit for bit against the oracle (itself checked against the reference's manifold walk on the CPU)
        for k in (0, 1):
            util.assert_same_contacts(util.records_to_contacts(eng.all_vs_all_poses(sets[k])), w.all_vs_all(sets[k]), "all-vs-all set %d" % k)
    w.close()


def _assert_same_sweeps(rec, hits, tag):
    """GPU records vs oracle hits, BIT FOR BIT.  The oracle runs in portable-trig mode (set_trig_mode(1)): the device
    cannot call glibc's sinf / cosf / acosf and the advancement is chaotically sensitive to their last bit, so both
    sides use the same fp64-series routines here; in libm mode the oracle is bit-exact against the reference's Bullet
    (tests/test_oracle_cpu.py)."""
    rk = [(int(r["t"]), int(r["idA"]), int(r["childA"])) for r in rec]
    hk = [(int(h["seg"]), int(h["obj"]), int(h["child"])) for h in hits]
    assert sorted(rk) == sorted(hk), "%s: hit sets differ: %s" % (tag, sorted(set(rk) ^ set(hk))[:8])
    byk = {k: h for k, h in zip(hk, hits)}
    exact = 0
    for k, r in zip(rk, rec):
        h = byk[k]
        assert np.float32(r["time"]) == h["fraction"], (tag, k, r["time"], h["fraction"])
        assert (int(r["idB"]), int(r["childB"])) == (int(h["hitObj"]), int(h["hitChild"])), (tag, k)
        assert np.array_equal(r["n"], h["n"]) and np.array_equal(r["ptA"], h["pt"]), (tag, k, r["n"], h["n"], r["ptA"], h["pt"])
        assert np.array_equal(r["ptA"], r["ptB"]) and r["dist"] == 0
    return len(rk)


def test_reference_continuous_check_known_answer(eng):
    # collision-checker-unit.cpp:102-120: boxbot swept from (-1.9, 0) to (0, 1.9) past the unit box -> one collision,
    # robot normal (-1, 0, 0)
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    bot = s.add_object([(b, M.IDENT7)], M.GROUP_ROBOT)
    s.finalize()
    eng.upload_scene(s, [T(0), T(0)])
    eng.set_contact_distance(0.05)
    rec = eng.continuous_check_poses([[T(0), T(-1.9, 0)], [T(0), T(0, 1.9)]], [bot])
    assert len(rec) == 1
    robot_normal = (1 if rec[0]["idA"] == bot else -1) * rec[0]["n"]
    assert np.allclose(robot_normal, (-1, 0, 0), atol=1e-4)
    assert np.float32(rec[0]["time"]) == np.float32(0.4736842)     # the reference Bullet's own fraction (oracle/_ref)
    # through the robot model: boxbot has the affine DOFs x, y (cast-cost-unit.cpp:44, RobotAndDOF(boxbot, [], DOF_X|DOF_Y))
    s2 = M.Scene()
    b2 = s2.add_box(.5, .5, .5)
    s2.add_object([(b2, M.IDENT7)], M.GROUP_KINBODY)
    rb = synth.boxbot(s2)
    s2.finalize(); rb.finalize()
    eng.upload_scene(s2, np.tile(np.array(M.IDENT7), (s2.n_objects, 1)))
    eng.upload_robot(rb)
    rec2 = eng.continuous_check_trajectory(np.array([[-1.9, 0.0], [0.0, 1.9], [1.9, 3.8]]), 1, 3)
    assert len(rec2) == 1 and rec2[0]["t"] == 0 and np.float32(rec2[0]["time"]) == np.float32(0.4736842)
    # a trajectory that stays clear reports nothing (what cast-cost-unit.cpp:75-77 asserts after optimisation)
    assert len(eng.continuous_check_trajectory(np.array([[-1.9, 0.0], [-1.9, 1.9], [1.9, 3.8]]), 1, 3)) == 0


@pytest.mark.parametrize("seed,spread", [(500, 0.8), (501, 1.5), (502, 0.4), (503, 1.5), (504, 0.8), (505, 0.6), (950, 0.6), (951, 0.8)])
def test_continuous_check_random_scenes_vs_oracle(eng, seed, spread):
    """ContinuousCheckTrajectory on explicit poses (boxes, hulls, cylinders, compounds; seeds >= 900: spheres) against the
    CPU oracle, which is bit-exact against the reference's convexSweepTest (tests/test_oracle_cpu.py)."""
    rng = np.random.default_rng(seed)
    s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=5, round_prob=0.8 if seed >= 900 else (0.3 if seed % 2 else 0.0), spheres=seed >= 900)
    nst = 5
    traj = np.stack([np.array([synth.random_T7(rng, -spread, spread) for _ in robot]) for _ in range(nst)])
    sets = np.tile(objT[None], (nst, 1, 1))
    sets[:, robot] = traj
    eng.upload_scene(s, objT)
    w = B.oracle_world(s)
    w.set_trig_mode(1)
    total = 0
    try:
        for d in (0.0, 0.05):
            eng.set_contact_distance(d)
            w.set_contact_distance(d)
            rec = eng.continuous_check_poses(sets, robot)
            hits = w.continuous_check(objT, robot, traj)
            total += _assert_same_sweeps(rec, hits, "seed %d d=%g" % (seed, d))
    finally:
        w.set_trig_mode(0)
        w.close()
    assert total > 0


def _arm_world(active, boxes, seed=0):
    s = M.Scene()
    rb = synth.pr2_like(s, seed=seed, active=active)
    ids, boxT = synth.add_boxes(s, boxes)
    s.finalize(); rb.finalize()
    static_T = np.tile(np.array(M.IDENT7), (s.n_objects, 1))
    static_T[ids] = boxT
    return s, rb, static_T


TABLE = [((.85, .55, .01), synth.T7((1.1, 0, .635)))]  # data/table.xml


@pytest.mark.parametrize("active,B_,T_", [("rightarm", 3, 10), ("fullbody", 2, 8)])
def test_robot_linearization_vs_oracle(eng, active, B_, T_):
    """FK + broadphase + narrowphase + Jacobian epilogue for B problems x T timesteps (configs 1-3 shape)."""
    s, rb, static_T = _arm_world(active, TABLE + [((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))])
    rng = np.random.default_rng(11)
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    w = B.oracle_world(s)
    trajs = []
    for b in range(B_):
        q0 = rng.uniform(-0.6, 0.6, size=rb.K)
        q1 = q0 + rng.uniform(-0.9, 0.9, size=rb.K)
        if active == "rightarm":
            q0 = rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, size=7); q1 = q0 + rng.uniform(-0.9, 0.9, size=7)
        trajs.append(synth.straight_line(q0, q1, T_))
    dofs = np.stack(trajs)
    # device FK against the oracle's FK
    fk = eng.forward_kinematics(dofs.reshape(-1, rb.K))
    for i, q in enumerate(dofs.reshape(-1, rb.K)):
        assert np.abs(fk[i] - B.oracle_fk(rb, q)).max() < 1e-13
    for d in (0.065, 0.09):
        eng.set_contact_distance(d)
        w.set_contact_distance(d)
        for mask in (-1, M.GROUP_KINBODY):
            rec, g = eng.links_vs_all_batch(dofs, B_, T_, filter_mask=mask)
            for b in range(B_):
                orec, og = w.linearize(rb, static_T, dofs[b], mask=mask)
                sel = rec["problem"] == b
                util.assert_same_linearization(rec[sel], g[sel], orec, og, "disc b=%d d=%g mask=%d" % (b, d, mask))
        rec, g = eng.cast_vs_all_batch(dofs, B_, T_, gap=1)
        ntot = 0
        for b in range(B_):
            orec, og = w.linearize(rb, static_T, dofs[b], cast=True)
            sel = rec["problem"] == b
            util.assert_same_linearization(rec[sel], g[sel], orec, og, "cast b=%d d=%g" % (b, d))
            ntot += len(orec)
        assert ntot > 0
    # want_jacobians=0 (CalcDists path) returns the same contacts
    rec0, g0 = eng.links_vs_all_batch(dofs, B_, T_, want_jacobians=False)
    rec1, _ = eng.links_vs_all_batch(dofs, B_, T_)
    assert g0 is None and np.array_equal(rec0["dist"], rec1["dist"]) and np.array_equal(rec0["idA"], rec1["idA"])
    w.close()


def test_robot_vs_spheres_and_cylinders(eng):
    """The robot path (FK, broadphase pre-test with margins, GJK / EPA, Jacobian epilogue) with sphere and cylinder
    obstacles (GT_Sphere / GT_Cylinder, bullet_collision_checker.cpp:150-158), bare and as children of a compound,
    and a sphere attached to a robot link's compound -- discrete (sphere-sphere analytic included) and cast."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    rng = np.random.default_rng(21)
    ids, Ts = [], []
    for i in range(10):
        ids.append(s.add_object([(s.add_sphere(rng.uniform(0.04, 0.15)), M.IDENT7)], M.GROUP_KINBODY, name="ball%d" % i))
        Ts.append(synth.T7(rng.uniform([0.3, -0.8, 0.3], [1.0, 0.3, 1.3])))
    ids.append(s.add_object([(s.add_sphere(0.08), synth.T7((0.1, 0, 0))), (s.add_cylinder(0.05, 0.2), synth.T7((-0.1, 0, 0), (1, 0, 0), 0.7)),
                             (s.add_sphere(0.05), synth.T7((0, 0.12, 0)))], M.GROUP_KINBODY, compound=True, name="cluster"))
    Ts.append(synth.T7((0.7, -0.3, 0.9)))
    s.finalize(); rb.finalize()
    static_T = synth.static_poses(s, ids, np.array(Ts))
    B_, T_ = 3, 8
    dofs = np.stack([synth.straight_line(rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, 7), rb.q_robot[1:8] + rng.uniform(-1.0, 1.0, 7), T_) for _ in range(B_)])
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_contact_distance(0.09)
    w = B.oracle_world(s)
    w.set_contact_distance(0.09)
    n = 0
    for cast in (False, True):
        rec, g = eng.cast_vs_all_batch(dofs, B_, T_, 1) if cast else eng.links_vs_all_batch(dofs, B_, T_)
        for b in range(B_):
            orec, og = w.linearize(rb, static_T, dofs[b], cast=cast)
            sel = rec["problem"] == b
            util.assert_same_linearization(rec[sel], g[sel], orec, og, "spheres cast=%s b=%d" % (cast, b))
            n += len(orec)
    assert n > 50
    w.close()


def test_grabbed_bodies(eng):
    """robot->Grab(body, link) (configuration_space.cpp:120-131): the grabbed body's links are query links with the
    GRABBER's Jacobian, whether or not an active DOF moves the grabber -- here a box in the right gripper (moved by the
    active right arm) and a hull in the left gripper (left arm inactive: listed all the same, gradient rows all zero)."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    rng = np.random.default_rng(33)
    held_r = s.add_object([(s.add_box(0.03, 0.03, 0.12), M.IDENT7)], M.GROUP_KINBODY, name="held_r")
    held_l = s.add_object([(s.add_hull(synth.blob_hull(rng, scale=(0.05, 0.05, 0.05), nverts=20)), synth.T7((0.02, 0, 0))),
                           (s.add_sphere(0.04), synth.T7((0, 0, 0.08)))], M.GROUP_KINBODY, compound=True, name="held_l")
    gr = rb.grab(rb.link_name.index("r_gripper_palm_link"), synth.T7((0.12, 0, 0), (0, 1, 0), 0.3), held_r)
    gl = rb.grab(rb.link_name.index("l_gripper_palm_link"), synth.T7((0.12, 0, 0)), held_l)
    ids, boxT = synth.add_boxes(s, TABLE + [((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5)), ((.1, .1, .02), M.IDENT7), ((.1, .1, .02), M.IDENT7)])
    s.finalize(); rb.finalize()
    static_T = synth.static_poses(s, ids, boxT)
    objs, links = rb.affected_links()
    assert held_r in objs and held_l in objs and links[objs.index(held_l)] == gl
    B_, T_ = 2, 8
    dofs = np.stack([synth.straight_line(rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, 7), rb.q_robot[1:8] + rng.uniform(-1.0, 1.0, 7), T_) for _ in range(B_)])
    # the last two boxes sit next to where the held objects are (FK of the build's own model)
    fk = B.oracle_fk(rb, dofs[0, T_ // 2])
    static_T[ids[-2], 4:] = fk[gr, 4:] + [0.0, 0.0, -0.1]
    static_T[ids[-1], 4:] = fk[gl, 4:] + [0.05, 0.0, -0.1]
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_contact_distance(0.09)
    w = B.oracle_world(s)
    w.set_contact_distance(0.09)
    seen_r = seen_l = 0
    for cast in (False, True):
        rec, g = eng.cast_vs_all_batch(dofs, B_, T_, 1) if cast else eng.links_vs_all_batch(dofs, B_, T_)
        for b in range(B_):
            orec, og = w.linearize(rb, static_T, dofs[b], cast=cast)
            sel = rec["problem"] == b
            util.assert_same_linearization(rec[sel], g[sel], orec, og, "grabbed cast=%s b=%d" % (cast, b))
        # the held objects as the query side (B after the cast swap; either side in the discrete pass)
        env = np.isin(rec["idA"], ids) | np.isin(rec["idB"], ids)
        mr = env & ((rec["idA"] == held_r) | (rec["idB"] == held_r))
        ml = env & ((rec["idA"] == held_l) | (rec["idB"] == held_l))
        seen_r += int(mr.sum()); seen_l += int(ml.sum())
        assert np.abs(g[mr]).max(initial=0) > 0 or not mr.any()
        assert not g[ml].any()            # nothing active moves the left gripper: d dist / d q = 0, the contact is still reported
    assert seen_r > 0 and seen_l > 0
    w.close()


def test_device_fk_and_jacobian_rows_closed_forms(eng):
    """The gradient half of the metric against forms that do not come from the build's own FK model (OpenRAVE, which would
    pin it, is absent -- DESIGN.md 2): (i) data/three_links.robot.xml, a planar 3R arm: k_fk's link poses against the
    closed-form chain, k_finish's rows against n . (z x (p - o_j)) with the joint origins o_j in closed form; (ii) a free
    rigid body driven by X | Y | Z | rotation about z: rows against [I | skew] (src/sandbox/incremental_rb.hpp:45-54)."""
    # (i) planar 3R
    s = M.Scene()
    rb = synth.three_links(s)
    ids, boxT = synth.add_boxes(s, [((.05, .05, .05), synth.T7((0.30, 0.12, 0.0))), ((.04, .2, .05), synth.T7((0.12, 0.20, 0.0), (0, 0, 1), 0.4))])
    s.finalize(); rb.finalize()
    static_T = synth.static_poses(s, ids, boxT)
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    rng = np.random.default_rng(5)
    Q = rng.uniform(-1.2, 1.2, size=(6, 3))
    fk = eng.forward_kinematics(Q)
    L = 0.16
    for q, Tl in zip(Q, fk):
        a1, a2, a3 = q[0], q[0] + q[1], q.sum()
        o1 = np.zeros(3); o2 = np.array([L * np.cos(a1), L * np.sin(a1), 0]); o3 = o2 + [L * np.cos(a2), L * np.sin(a2), 0]
        for link, o, ang in ((1, o1, a1), (2, o2, a2), (3, o3, a3)):
            assert np.allclose(Tl[link][4:], o, atol=1e-14)
            assert np.allclose(Tl[link][:4], [np.cos(ang / 2), 0, 0, np.sin(ang / 2)], atol=1e-14)
    eng.set_contact_distance(0.25)
    rec, g = eng.links_vs_all_batch(Q[:, None, :].reshape(6, 3), 6, 1, filter_mask=M.GROUP_KINBODY)
    assert len(rec) >= 6
    arm = {o: i for i, o in enumerate(rb.link_obj) if o >= 0}     # object -> link index (1, 2, 3)
    z = np.array([0.0, 0.0, 1.0])
    for r, row in zip(rec, g[:, 0]):
        q = Q[r["problem"]]
        a1, a2 = q[0], q[0] + q[1]
        origins = [np.zeros(3), np.array([L * np.cos(a1), L * np.sin(a1), 0])]
        origins.append(origins[1] + [L * np.cos(a2), L * np.sin(a2), 0])
        want = np.zeros(3)
        for obj, pt, sign in ((r["idA"], r["ptA"], 1.0), (r["idB"], r["ptB"], -1.0)):   # collision_terms.cpp:44-56
            if obj in arm:
                for j in range(arm[obj]):                                                # joints 0 .. link-1 move the link
                    want[j] += sign * r["n"] @ np.cross(z, pt - origins[j])
        assert np.allclose(row, want, rtol=1e-12, atol=1e-14)
        assert abs(r["const0"] + row @ q - r["dist"]) < 1e-13
    # (ii) free rigid body
    s2 = M.Scene()
    bot = synth.boxbot(s2)
    bot.set_active([], M.DOF_X | M.DOF_Y | M.DOF_Z | M.DOF_ROTAXIS, (0, 0, 1))
    ids2, boxT2 = synth.add_boxes(s2, [((.5, .5, .5), M.IDENT7)])
    s2.finalize(); bot.finalize()
    st2 = synth.static_poses(s2, ids2, boxT2)
    eng.upload_scene(s2, st2)
    eng.upload_robot(bot)
    eng.set_contact_distance(0.3)
    X = np.array([[1.15, 0.2, 0.1, 0.3], [-0.1, 1.2, -0.2, -0.6], [0.3, -0.2, 1.1, 1.0]])
    rec, g = eng.links_vs_all_batch(X, 3, 1)
    assert len(rec) == 3
    for r, row in zip(rec, g[:, 0]):
        x = X[r["problem"]]
        sign, pt = (1.0, r["ptA"]) if r["idA"] == bot.link_obj[0] else (-1.0, r["ptB"])
        J = np.concatenate([np.eye(3), np.cross(z, pt - x[:3])[:, None]], axis=1)        # [I | z x (p - o)]
        assert np.allclose(row, sign * r["n"] @ J, rtol=1e-12, atol=1e-14)


def test_hinge_block_equals_host_assembly(eng):
    """tohc_fetch_hinges -- the hinge rows and BPMPD columns assembled on the device -- against the same
    block built on the host from the fetched contact slab the way sco does it (collision_checker.bpmpd_collision_block:
    CollisionsToDistanceExpressions -> convex -> addHinge -> BPMPDModel::optimize with simplify2): BIT-identical, for a
    discrete + cast slab of several problems, a cast pass with gap 2, a discrete-only slab, an empty slab; and the
    begin / end pair overlapping the next linearization returns the same block as the synchronous call."""
    from trajopt_amd import collision_checker as CC, engine
    s, rb, static_T = _arm_world("fullbody", TABLE + [((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))])
    rng = np.random.default_rng(17)
    B_, T_ = 3, 9
    dofs = np.stack([synth.straight_line(rng.uniform(-0.6, 0.6, rb.K), rng.uniform(-0.9, 0.9, rb.K), T_) for _ in range(B_)])
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_contact_distance(0.12)
    eng.stage_dofs(dofs, B_, T_)

    def same(h, ref, tag):
        for k in ("B", "T", "K", "gap", "n_rows", "nnz"):
            assert h[k] == ref[k], (tag, k, h[k], ref[k])
        for k in ("row_off", "rhs", "dist", "row_t", "col_off", "col_idx", "col_val"):
            assert h[k].dtype == ref[k].dtype and h[k].tobytes() == ref[k].tobytes(), (tag, k)

    for gap, passes in ((1, "dc"), (2, "dc"), (0, "d"), (1, "c")):
        first = True
        for kind in passes:
            if kind == "d":
                eng.enqueue_links_vs_all(-1, True, append=not first)
            else:
                eng.enqueue_cast_vs_all(gap, True, append=not first)
            first = False
        rec, g = eng.fetch()
        assert len(rec) > 50
        h = eng.fetch_hinges(0.05, 0.03)
        ref = CC.bpmpd_collision_block(rec, g, B_, T_, rb.K, gap if "c" in passes else 0, 0.05, 0.03)
        same(h, ref, "gap %d %s" % (gap, passes))
        assert h["nnz"] < h["n_rows"] * 2 * rb.K          # structural zeros (joints that do not move the link) are dropped
        # the rows say what the reference's costs say: hinge argument at x = dist_pen - dist
        x = dofs.reshape(B_, -1)
        for p_ in range(B_):
            r0, r1 = h["row_off"][p_], h["row_off"][p_ + 1]
            viol = -h["rhs"][r0:r1].copy()
            for c in range(T_ * rb.K):
                a, b = h["col_off"][p_ * T_ * rb.K + c], h["col_off"][p_ * T_ * rb.K + c + 1]
                np.add.at(viol, h["col_idx"][a:b], h["col_val"][a:b] * x[p_, c])
            pen = np.where(h["row_t"][r0:r1] & engine.HINGE_CAST, 0.03, 0.05)
            assert np.allclose(viol, pen - h["dist"][r0:r1], atol=1e-12)
    # begin / end around the next linearization
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    want = eng.fetch_hinges(0.05, 0.05)
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    eng.fetch_hinges_begin(0.05, 0.05)
    with pytest.raises(engine.EngineError):
        eng.fetch_hinges_begin(0.05, 0.05)               # one begin at a time
    eng.stage_dofs(dofs[::-1].copy(), B_, T_)            # the next batch: other trajectories
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    got = eng.fetch_hinges_end()
    same(got, want, "pipelined")
    nxt = eng.fetch_hinges(0.05, 0.05)
    assert nxt["n_rows"] == want["n_rows"] and np.array_equal(nxt["row_off"], np.concatenate([[0], np.cumsum(np.diff(want["row_off"])[::-1])]))
    with pytest.raises(engine.EngineError):
        eng.fetch_hinges_end()
    # an empty slab (nothing within the contact distance)
    eng.set_contact_distance(0.0)
    far = dofs.copy(); far[..., -3] += 50.0               # the base 50 m away
    eng.stage_dofs(far, B_, T_)
    eng.enqueue_links_vs_all(M.GROUP_KINBODY, True, append=False)
    h0 = eng.fetch_hinges(0.05)
    assert h0["n_rows"] == 0 and h0["nnz"] == 0 and not h0["col_off"].any() and not h0["row_off"].any()
    # a slab without gradient rows cannot be turned into hinges
    eng.enqueue_links_vs_all(-1, False, append=False)
    with pytest.raises(engine.EngineError):
        eng.fetch_hinges(0.05)


def test_allowed_matrix_and_group64_agree(eng):
    s, rb, static_T = _arm_world("rightarm", TABLE)
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_contact_distance(0.1)
    w = B.oracle_world(s)
    w.set_contact_distance(0.1)
    q = rb.q_robot[1:8][None]
    # IgnoreZeroStateSelfCollisions-style exclusion: forbid every pair found in this pose, both directions
    rec, _ = eng.links_vs_all_batch(q, 1, 1)
    assert len(rec) > 0
    m = np.ones((s.n_objects, s.n_objects), np.uint8)
    m[rec["idA"], rec["idB"]] = 0
    m[rec["idB"], rec["idA"]] = 0
    eng.set_allowed_matrix(m)
    w.set_allowed(m)
    rec2, g2 = eng.links_vs_all_batch(q, 1, 1)
    assert len(rec2) == 0
    q2 = q + 0.4
    rec3, g3 = eng.links_vs_all_batch(q2, 1, 1)
    orec, og = w.linearize(rb, static_T, q2)
    util.assert_same_linearization(rec3, g3, orec, og, "after exclusion")
    w.close()


def test_empty_and_error_paths(eng):
    from trajopt_amd import engine
    s, rb, static_T = _arm_world("rightarm", TABLE)
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_contact_distance(0.0)
    far = static_T.copy()
    rec = eng.links_vs_all_poses(far, [])          # empty query
    assert len(rec) == 0
    with pytest.raises(engine.EngineError):
        eng.set_contact_distance(-1.0)
    bad = M.Scene()                                 # an unknown shape kind: rejected loudly, never approximated
    bad.add_object([(bad.add_sphere(0.1), M.IDENT7)], M.GROUP_KINBODY)
    bad.finalize()
    bad.arrays["kind"][0] = 7
    with pytest.raises(engine.EngineError, match="kind 7"):
        eng.upload_scene(bad, [M.IDENT7])
    eng.upload_scene(s, static_T)
    eng.upload_robot(rb)
    eng.set_capacity(16, 4)                         # too small on purpose: reported, not silently truncated
    eng.set_contact_distance(0.2)
    with pytest.raises(engine.EngineError):
        eng.links_vs_all_batch(rb.q_robot[1:8][None], 1, 1)
    eng.set_capacity(1 << 22, 1 << 18)
    rec, _ = eng.links_vs_all_batch(rb.q_robot[1:8][None], 1, 1)
    assert len(rec) > 0


def test_engine_state_survives_failed_calls(built):
    """A fetch before any enqueue is an empty slab; a rejected upload leaves the previous scene and robot in place
    (same answers as before it); after a capacity overflow the truncated slab cannot be fetched or appended to; an
    appended pass must agree with the slab on want_jacobians."""
    from trajopt_amd import engine
    e = engine.Engine(0)
    try:
        rec, _ = e.fetch()
        assert len(rec) == 0
        s, rb, static_T = _arm_world("rightarm", TABLE)
        e.upload_scene(s, static_T)
        e.upload_robot(rb)
        e.set_contact_distance(0.2)
        q = rb.q_robot[1:8][None]
        ref, gref = e.links_vs_all_batch(q, 1, 1)
        assert len(ref) > 0
        # a second, larger scene that fails validation at its LAST object (the shapes and children before it are fine)
        big = M.Scene()
        for i in range(s.n_objects + 5):
            big.add_object([(big.add_box(.1, .1, .1), M.IDENT7)], M.GROUP_KINBODY)
        big.finalize()
        big.arrays["obj_cnt"][-1] = 3                   # child range runs past the end
        with pytest.raises(engine.EngineError, match="child range"):
            e.upload_scene(big, np.tile(np.array(M.IDENT7), (big.n_objects, 1)))
        bad = synth.pr2_like(M.Scene(), seed=0, active="rightarm")
        bad.finalize()
        bad.arrays["dof"][-1] = 99                      # out of range, found after the other links were read
        with pytest.raises(engine.EngineError, match="out of range"):
            e.upload_robot(bad)
        rec, g = e.links_vs_all_batch(q, 1, 1)          # the first scene and robot still answer, identically
        assert rec.tobytes() == ref.tobytes() and np.array_equal(g, gref)
        # overflow: sync reports it, and the slab it truncated is not handed out afterwards
        e.set_capacity(16, 4)
        e.stage_dofs(q, 1, 1)
        e.enqueue_links_vs_all(-1, True, append=False)
        with pytest.raises(engine.EngineError) as ex:
            e.sync()
        assert ex.value.code == -3
        with pytest.raises(engine.EngineError) as ex:
            e.fetch()
        assert ex.value.code == -4
        with pytest.raises(engine.EngineError) as ex:
            e.enqueue_cast_vs_all(1, True, append=True)
        assert ex.value.code == -4
        e.set_capacity(1 << 22, 1 << 18)
        e.enqueue_links_vs_all(-1, True, append=False)
        with pytest.raises(engine.EngineError) as ex:   # one slab, one layout
            e.enqueue_cast_vs_all(1, False, append=True)
        assert ex.value.code == -1
        e.enqueue_cast_vs_all(1, True, append=True)
        rec, g = e.fetch()
        disc = rec[rec["cctype"] == 0]
        assert disc.tobytes() == ref.tobytes() and g.shape == (len(rec), 2, rb.K)
    finally:
        e.close()


# ---- RayCastCollision (bullet_collision_checker.cpp:352-360) ---------------------------------------
def test_ray_cast_known_answers(eng):
    from tests.test_oracle_cpu import RAY_CASES   # fractions returned by the reference's own Bullet
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    s.finalize()
    eng.upload_scene(s, [T(0)])
    fr = eng.ray_cast_poses([T(0)], [c[0] for c in RAY_CASES], [c[1] for c in RAY_CASES])
    assert fr.tolist() == [float(np.float32(c[2])) for c in RAY_CASES]
    assert len(eng.ray_cast_poses([T(0)], np.zeros((0, 3)), np.zeros((0, 3)))) == 0       # no rays
    with pytest.raises(ValueError):
        eng.ray_cast_poses([T(0)], np.zeros((2, 3)), np.zeros((3, 3)))
    # the box moved out of the way: the poses of the call count, not those of upload_scene
    assert eng.ray_cast_poses([T(0, 5)], [(-2, 0, 0)], [(2, 0, 0)])[0] == 1.0


def test_ray_cast_golden_fixtures_bit_exact(eng):
    """closest hit fractions recorded from the reference's btCollisionWorld::rayTest (tests/golden/make_rays.py)"""
    zr = np.load(os.path.join(ROOT, "tests", "golden", "rays.npz"), allow_pickle=False)
    hits = 0
    for path in GOLDEN:
        z = np.load(path, allow_pickle=False)
        tag = os.path.basename(path)[len("random_scene_"):-4]
        s = M.scene_from_arrays({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
        eng.upload_scene(s, z["objT"])
        for i, d in enumerate((0.0, 0.05)):
            eng.set_contact_distance(d)
            fr = eng.ray_cast_poses(z["objT"], zr["from_" + tag], zr["to_" + tag])
            assert np.array_equal(fr.view(np.uint32), zr["frac_%s_%d" % (tag, i)].view(np.uint32)), "scene %s d=%g" % (tag, d)
            hits += int((fr < 1).sum())
    assert hits > 1000


@pytest.mark.parametrize("seed,spread", [(600, 0.8), (601, 1.5), (602, 0.4), (603, 0.6), (960, 0.6)])
def test_ray_cast_random_scenes_vs_oracle(eng, seed, spread):
    """boxes, hulls, z-cylinders (odd seeds), spheres (seed 960), compounds: bit-exact against the oracle, which is
    bit-exact against the reference's rayTest (tests/test_oracle_cpu.py)"""
    rng = np.random.default_rng(seed)
    s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=5, round_prob=0.8 if seed >= 900 else (0.3 if seed % 2 else 0.0), spheres=seed >= 900)
    n = 2000
    a = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    b = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    a[:50] = objT[rng.integers(0, len(objT), 50), 4:7]
    b[50:60] = a[50:60]
    eng.upload_scene(s, objT)
    w = B.oracle_world(s)
    fr, fo = eng.ray_cast_poses(objT, a, b), w.ray_cast(objT, a, b)
    w.close()
    assert (fo < 1).sum() > 100 and (fo == 1).sum() > 100
    assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32))


def test_ray_cast_fullsize_soup_through_checker(built):
    """Config 4's world (full-body robot + 4096-hull soup) through the host mirror of the reference interface:
    20000 rays at one configuration.  Size-independent properties + a sample against the oracle."""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_fullbody_soup()
    cc = CC.CreateCollisionChecker(s, static_T)
    cc.SetRobot(rb, ignore_zero_state=False)
    cc.SetContactDistance(d)
    rng = np.random.default_rng(7)
    q = synth.fullbody_trajectory(rb, rng, 100)[37]
    cc.SetDOFValues(q)
    n = 20000
    a = np.column_stack([rng.uniform(-0.5, 4.0, n), rng.uniform(-1.8, 1.8, n), rng.uniform(1.5, 2.5, n)])
    b = np.column_stack([a[:, 0] + rng.uniform(-0.5, 0.5, n), a[:, 1] + rng.uniform(-0.5, 0.5, n), rng.uniform(-1.0, 0.2, n)])
    fr = cc.RayCastCollisions(a, b)
    assert (fr < 1).sum() > n // 10 and (fr == 1).sum() > n // 100 and (fr >= 0).all()
    assert np.array_equal(fr, cc.RayCastCollisions(a, b))                        # idempotent (shared scratch reset per call)
    assert np.array_equal(fr[5000:6000], cc.RayCastCollisions(a[5000:6000], b[5000:6000]))   # rays are independent
    for i in (int(np.argmax(fr < 1)), int(np.argmax(fr == 1))):                  # the reference's one-ray bool form
        assert cc.RayCastCollision(a[i], b[i]) == bool(fr[i] < 1)
    # a hit through the robot itself: from above the base link straight down through the robot
    fk = cc.eng.forward_kinematics(q[None])[0]
    objT = static_T.copy()
    for i, o in enumerate(rb.link_obj):
        if o >= 0:
            objT[o] = fk[i]
    w = B.oracle_world(s, cap=16)
    idx = rng.choice(n, 300, replace=False)
    fo = w.ray_cast(objT, a[idx], b[idx])
    assert np.array_equal(fr[idx].view(np.uint32), fo.view(np.uint32))
    # robot links take part (RayCastCollision has no group filter): rays aimed at link origins from 1 m above
    tgt = np.array([objT[o, 4:7] for o in rb.link_obj if o >= 0])
    src = tgt + np.array([0.0, 0.0, 1.0])
    fr2 = cc.RayCastCollisions(src, tgt - np.array([0.0, 0.0, 0.5]))
    fo2 = w.ray_cast(objT, src, tgt - np.array([0.0, 0.0, 0.5]))
    w.close()
    assert np.array_equal(fr2.view(np.uint32), fo2.view(np.uint32)) and (fr2 < 1).sum() >= len(tgt) // 2
    cc.eng.close()


def test_fullsize_soup_properties(built):
    """Config 4 (full body, 100 timesteps, 4096-hull soup, zero-state self-collision exclusions like the reference's
    AddKinBody path): too big for the scalar oracle in the test budget, so check size-independent properties and
    spot-check single timesteps against the oracle.  Goes through the host mirror of the reference interface."""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_fullbody_soup()
    cc = CC.CreateCollisionChecker(s, static_T)
    cc.SetRobot(rb)                  # + IgnoreZeroStateSelfCollisions at d = 0.1
    cc.SetContactDistance(d)
    eng = cc.eng
    assert (cc.allowed == 0).sum() > 0 and (cc.allowed == cc.allowed.T).all()
    rng = np.random.default_rng(2)
    traj = synth.fullbody_trajectory(rb, rng, 100)
    rec, g = eng.links_vs_all_batch(traj, 1, 100)
    c = eng.counters()
    assert c["contacts"] == len(rec) > 0 and c["pair_evals"] >= len(rec)
    # determinism / idempotence: a second call returns the identical slab
    rec2, g2 = eng.links_vs_all_batch(traj, 1, 100)
    assert np.array_equal(rec, rec2) and np.array_equal(g, g2)
    # every record obeys the contact filter and the affine identity const0 + grad0.q = dist
    assert (rec["dist"] <= np.float32(d)).all()
    assert np.allclose(rec["const0"] + np.einsum("nk,nk->n", g[:, 0], traj[rec["t"]]), rec["dist"], atol=1e-9)
    assert np.allclose(np.linalg.norm(rec["n"], axis=1), 1, atol=1e-5)
    # timestep-sharding invariance (what multi-GPU sharding by timestep relies on)
    sub, gs = eng.links_vs_all_batch(traj[40:50], 1, 10)
    sel = (rec["t"] >= 40) & (rec["t"] < 50)
    assert np.array_equal(sub["dist"], rec[sel]["dist"]) and np.array_equal(sub["idB"], rec[sel]["idB"]) and np.array_equal(gs, g[sel])
    # resident API: discrete + cast appended into one slab == the two one-shot calls
    recc, gc = eng.cast_vs_all_batch(traj, 1, 100, 1)
    eng.stage_dofs(traj, 1, 100)
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    both, gboth = eng.fetch()
    assert len(both) == len(rec) + len(recc)
    disc = both["cctype"] == 0
    assert np.array_equal(both[disc], rec) and np.array_equal(gboth[disc], g)
    assert np.array_equal(both[~disc], recc) and np.array_equal(gboth[~disc], gc)
    cnt = eng.counters()
    assert cnt["pair_evals"] > 0 and cnt["cum_ms_gjk"] > 0
    # spot check single timesteps against the oracle, discrete and cast
    w = B.oracle_world(s, cap=1 << 17)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    for t in (0, 57, 98):
        orec, og = w.linearize(rb, static_T, traj[t][None])
        r = rec[rec["t"] == t].copy()
        r["t"] = 0
        util.assert_same_linearization(r, g[rec["t"] == t], orec, og, "soup disc t=%d" % t)
        orec, og = w.linearize(rb, static_T, traj[t:t + 2], cast=True)
        r = recc[recc["t"] == t].copy()
        r["t"] = 0
        util.assert_same_linearization(r, gc[recc["t"] == t], orec, og, "soup cast t=%d" % t)
    # ContinuousCheckTrajectory of the whole trajectory: determinism, one hit at most per (segment, link geometry),
    # fractions in [0, 1], unit normals, KinBody objects only; spot check of two segments against the oracle (which has
    # no cull: every one of the 4096 soup hulls goes through the advancement) for a handful of links
    sw = eng.continuous_check_trajectory(traj, 1, 100)
    assert np.array_equal(sw, eng.continuous_check_trajectory(traj, 1, 100))
    keys = list(zip(sw["t"].tolist(), sw["idA"].tolist(), sw["childA"].tolist()))
    assert len(sw) > 0 and len(set(keys)) == len(keys)
    assert (sw["time"] >= 0).all() and (sw["time"] <= 1).all() and (sw["t"] < 99).all()
    assert np.allclose(np.linalg.norm(sw["n"], axis=1), 1, atol=1e-5)
    assert (np.asarray(s.arrays["obj_group"])[sw["idB"]] & M.GROUP_KINBODY).all()
    # a trajectory sharded by timestep (with the one-waypoint halo) gives the same hits
    sub = eng.continuous_check_trajectory(traj[30:41], 1, 11)
    sel = (sw["t"] >= 30) & (sw["t"] < 40)
    assert np.array_equal(sub["time"], sw[sel]["time"]) and np.array_equal(sub["idB"], sw[sel]["idB"]) and np.array_equal(sub["n"], sw[sel]["n"])
    fk = eng.forward_kinematics(traj)
    links = [i for i, o in enumerate(rb.link_obj) if o >= 0]
    hit_links = sorted(set(sw["idA"].tolist()))
    objs = (hit_links[:3] + [o for o in (rb.link_obj[i] for i in links) if o not in hit_links][:2])
    lk = [list(rb.link_obj).index(o) for o in objs]
    w.set_trig_mode(1)
    try:
        for seg in (int(sw["t"][0]), int(sw["t"][len(sw) // 2])):
            hits = w.continuous_check(static_T, objs, fk[seg:seg + 2][:, lk])
            mine = sw[(sw["t"] == seg) & np.isin(sw["idA"], objs)].copy()
            mine["t"] = 0
            _assert_same_sweeps(mine, hits, "soup sweep seg %d" % seg)
    finally:
        w.set_trig_mode(0)
    w.close()
    eng.close()


def test_cast_cull_box_rounds_like_the_reference(built):
    """CastHullShape::getAabb (bullet_collision_checker.cpp:850-856) joins the box at t with the box at t * m_t01, and m_t01 =
    tf0^-1 * tf1 is rounded: t * m_t01 is tf1 to an ulp, not tf1.  With the box of tf1 itself the engine once missed one box
    overlap of 22.6 million on bench.py's workload (problem 18, cast unit 49: a pair evaluation the oracle counted; no
    contact hangs on it).  k_cast_boxes makes the second box the reference's way: the unit's pair evaluations equal the
    oracle's, and so do its records.  (scripts/check_bench_step.py checks the whole 64-problem step the same way.)"""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0)
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb); cc.SetContactDistance(d)
    eng = cc.eng
    traj = synth.fullbody_trajectory(rb, np.random.default_rng(1000 + 18), 100)
    w = B.oracle_world(s, cap=1 << 17)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    for t in (49, 48, 50):
        q = traj[t:t + 2]
        eng.stage_dofs(q[None], 1, 2)
        eng.enqueue_cast_vs_all(1, True, append=False)
        rec, g = eng.fetch()
        c0 = w.counter(0)
        orec, og = w.linearize(rb, static_T, q, cast=True)
        assert eng.counters()["pair_evals"] == w.counter(0) - c0, t
        util.assert_same_linearization(rec, g, orec, og, "cast unit %d of problem 18" % t)
    w.close()
    eng.close()


def test_bench_workload_whole_problems_vs_oracle(built):
    """bench.py's default workload (config 4 as BASELINE.md has it: 4096 hulls of 42 vertices), three whole problems of 100
    timesteps in one call, discrete + cast with gradient rows: EVERY record against the oracle and the pair-evaluation count
    of the headline metric equal to the oracle's.  (scripts/check_bench_step.py does this for all 64 problems of the timed
    step on 14 cores; profiles/r03_check_*.json.)"""
    import bench
    s, rb, static_T, d, cc, dofs = bench.build_world("fullbody42", 0, 1, 3)
    eng = cc.eng
    eng.stage_dofs(dofs, 3, 100)
    eng.enqueue_linearization(-1, 1, True)
    rec, g = eng.fetch()
    pe = eng.counters()["pair_evals"]
    w = B.oracle_world(s, cap=1 << 18)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    c0 = w.counter(0)
    for b in range(3):
        sel = rec["problem"] == b
        disc, cast = sel & (rec["cctype"] == 0), sel & (rec["cctype"] != 0)
        orec, og = w.linearize(rb, static_T, dofs[b])
        util.assert_same_linearization(rec[disc], g[disc], orec, og, "bench problem %d discrete" % b)
        orec, og = w.linearize(rb, static_T, dofs[b], cast=True)
        util.assert_same_linearization(rec[cast], g[cast], orec, og, "bench problem %d cast" % b)
    assert pe == w.counter(0) - c0 and len(rec) > 1000
    w.close()
    eng.close()
