"""CPU suite (-m "not gpu"): pins the oracle, the host logic and the C-ABI surface.

  * reference known answers (collision-checker-unit.cpp:48-87,130-150): contact counts, the
    -0.0577350222 diagonal-normal EPA answer, cast normals and times;
  * golden fixtures recorded from the reference's own vendored Bullet (tests/golden/make_golden.py): bit-exact;
  * where oracle/_ref/libref.so exists (build container), fresh random scenes against real Bullet: bit-exact;
  * the device narrowphase source, compiled for the host (tests/emu): bit-exact against the oracle;
  * FK / Jacobian: finite differences + closed-form planar arm (OpenRAVE parity is UNPINNED);
  * libtrajopt_hipcc.so loads and exports every symbol of include/trajopt_hipcc.h; no compute without a GPU.
"""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from oracle import bindings as B
from trajopt_amd import model as M, synth
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "random_scene_*.npz")))


def T(x, y=0, z=0):
    return (1, 0, 0, 0, x, y, z)


def two_boxes(group1):
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    o0 = s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY, name="box0")
    o1 = s.add_object([(b, M.IDENT7)], group1, name="box1")
    return s.finalize(), o0, o1


WORLDS = [B.oracle_world, util.emu_world] + ([B.ref_world] if B.ref_available() else [])


@pytest.mark.parametrize("mk", WORLDS)
def test_reference_box_distance_known_answers(built, mk):
    # collision-checker-unit.cpp:48-87
    s, o0, o1 = two_boxes(M.GROUP_KINBODY)
    w = mk(s)
    expect = [(0, .9, 1), (0, 1.1, 0), (.04, 1.1, 0), (.1, 1.09, 1), (.2, 1.1, 1)]
    for d, x, n in expect:
        w.set_contact_distance(d)
        c = w.links_vs_all([T(0), T(x)], [o0])
        assert len(c) == n, (d, x)
    # values observed from the reference's Bullet in the survey
    w.set_contact_distance(0)
    c = w.links_vs_all([T(0), T(.9)], [o0])[0]
    assert np.float32(c["dist"]) == np.float32(-0.0577350222)
    assert np.allclose(c["n"], [-0.57735, 0.57735, -0.57735], atol=1e-5)
    w.set_contact_distance(.1)
    c = w.links_vs_all([T(0), T(1.09)], [o0])[0]
    assert np.float32(c["dist"]) == np.float32(0.0900000334) and np.array_equal(c["n"], [-1, 0, 0])
    w.set_contact_distance(.2)
    c = w.links_vs_all([T(0), T(1.1)], [o0])[0]
    assert np.float32(c["dist"]) == np.float32(0.100000016) and np.array_equal(c["n"], [-1, 0, 0])
    w.close()


@pytest.mark.parametrize("mk", WORLDS)
def test_reference_cast_known_answers(built, mk):
    # collision-checker-unit.cpp:102-150 (CastVsAll part): boxbot swept past a box
    s, box, bot = two_boxes(M.GROUP_ROBOT)
    w = mk(s)
    cases = [((-1.9, 0), (0, 1.9), (-1 / np.sqrt(2), 1 / np.sqrt(2), 0), .5, 1e-1),
             ((0, .9), (0, 2), (0, 1, 0), 0, 1e-6), ((0, 2), (0, .9), (0, 1, 0), 1, 1e-6)]
    for a, b, nrm, time, ttol in cases:
        c = w.cast_vs_all([T(0), T(*a)], [bot], [T(*a)], [T(*b)])
        assert len(c) == 1
        robot_normal = (1 if c[0]["objA"] == bot else -1) * c[0]["n"]
        assert np.allclose(robot_normal, nrm, atol=1e-4)
        assert abs(c[0]["time"] - time) <= ttol
        assert c[0]["objB"] == bot  # the collector orients the contact so the swept link is B
    w.close()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("mk", [B.oracle_world, util.emu_world])
def test_golden_fixtures_bit_exact(built, path, mk):
    z = np.load(path, allow_pickle=False)
    s = M.scene_from_arrays({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
    w = mk(s)
    robot = z["robot"]
    for i, d in enumerate(z["dists"]):
        w.set_contact_distance(float(d))
        util.assert_same_contacts(w.links_vs_all(z["objT"], robot), z["disc_%d" % i], "disc d=%g" % d)
        util.assert_same_contacts(w.cast_vs_all(z["objT"], robot, z["objT"][robot], z["T1"]), z["cast_%d" % i], "cast d=%g" % d)
    w.close()


# ---- RayCastCollision (bullet_collision_checker.cpp:352-360) ---------------------------------------
RAY_CASES = [  # unit box at the origin; fractions are what the reference's own Bullet returns (oracle/_ref)
    ((-2, 0, 0), (2, 0, 0), 0.375),                 # straight through
    ((-2, 2, 0), (2, 2, 0), 1.0),                   # passes beside it
    ((0, 0, 0), (2, 0, 0), 1.0),                    # starts inside: btSubsimplexConvexCast reports nothing
    ((2, 0, 0), (3, 0, 0), 1.0),                    # points away
    ((-2, 0, 0), (-2, 0, 0), 1.0),                  # zero length
    ((-2, 0.3, 0.1), (2, -0.2, 0.3), 0.37474722),   # oblique: the cast stops within its 1e-4 squared-distance tolerance
    ((-2, 0, 0), (-1.6, 0, 0), 1.0),                # ends before the box
]


@pytest.mark.parametrize("mk", WORLDS)
def test_ray_cast_known_answers(built, mk):
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    s.finalize()
    w = mk(s)
    fr = w.ray_cast([T(0)], [c[0] for c in RAY_CASES], [c[1] for c in RAY_CASES])
    assert fr.tolist() == [float(np.float32(c[2])) for c in RAY_CASES]
    w.close()


@pytest.mark.parametrize("mk", [B.oracle_world, util.emu_world])
def test_ray_cast_golden_fixtures_bit_exact(built, mk):
    """tests/golden/rays.npz: closest hit fractions recorded from the reference's btCollisionWorld::rayTest
    (tests/golden/make_rays.py) on the scenes of random_scene_*.npz."""
    zr = np.load(os.path.join(ROOT, "tests", "golden", "rays.npz"), allow_pickle=False)
    hits = 0
    for path in GOLDEN:
        z = np.load(path, allow_pickle=False)
        tag = os.path.basename(path)[len("random_scene_"):-4]
        s = M.scene_from_arrays({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
        w = mk(s)
        for i, d in enumerate((0.0, 0.05)):
            w.set_contact_distance(d)
            fr = w.ray_cast(z["objT"], zr["from_" + tag], zr["to_" + tag])
            assert np.array_equal(fr.view(np.uint32), zr["frac_%s_%d" % (tag, i)].view(np.uint32)), "scene %s d=%g" % (tag, d)
            hits += int((fr < 1).sum())
        w.close()
    assert hits > 1000


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("seed", range(500, 512))
def test_ray_cast_oracle_vs_reference_bullet(built, seed):
    """The C restatement and the host build of the device code against the reference's rayTest, bit for bit; seeds
    3 mod 4 hold spheres."""
    rng = np.random.default_rng(seed)
    spread = float(rng.choice([0.4, 0.8, 1.5]))
    spheres = seed % 4 == 3
    s, objT, robot = synth.random_scene(rng, spread=spread, round_prob=0.3 if seed % 2 else 0.0, spheres=spheres)
    n = 300
    a = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    b = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    a[:20] = objT[rng.integers(0, len(objT), 20), 4:7]
    b[20:25] = a[20:25]
    worlds = [B.ref_world(s), B.oracle_world(s), util.emu_world(s)]
    for d in (0.0, 0.05):
        for w in worlds:
            w.set_contact_distance(d)
        fr = [w.ray_cast(objT, a, b) for w in worlds]
        assert (fr[0] < 1).any()
        for f in fr[1:]:
            assert np.array_equal(f.view(np.uint32), fr[0].view(np.uint32))
    for w in worlds:
        w.close()


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
def test_ray_cast_grazing_rays_and_the_reference_cull(built):
    """the restatement has no broadphase ray cull (Bullet's depends on the DBVT's history).  With the
    boxes grown by 3.001 d = 15 cm the cull never matters; with d = 0 the reference's tight boxes cull the rare hit
    that btSubsimplexConvexCast reports away from the shape, and the restatement reports it (never the other way round)."""
    rng = np.random.default_rng(705)
    spread = float(rng.choice([0.4, 0.8, 1.5]))
    s, objT, robot = synth.random_scene(rng, spread=spread, n_static=10, n_robot=6, round_prob=0.4)
    n = 20000
    a = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    b = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
    k = n // 2
    tgt = objT[rng.integers(0, len(objT), k), 4:7] + rng.normal(size=(k, 3)) * 0.15
    b[:k] = a[:k] + (tgt - a[:k]) * rng.uniform(0.5, 3.0, (k, 1))
    wr, wo = B.ref_world(s), B.oracle_world(s)
    seen = []
    # d = 0 first: queried after d = 0.05 the reference's DBVT still holds the larger boxes (btDbvt::update keeps a
    # volume that contains the new one) and culls nothing either -- its answer depends on that history
    for d in (0.0, 0.05, 0.0):
        wr.set_contact_distance(d); wo.set_contact_distance(d)
        fr, fo = wr.ray_cast(objT, a, b), wo.ray_cast(objT, a, b)
        ne = fr.view(np.uint32) != fo.view(np.uint32)
        seen.append(int(ne.sum()))
        assert (fo[ne] < fr[ne]).all()
    assert 0 < seen[0] <= 10 and seen[1] == 0 and seen[2] == 0
    wr.close(); wo.close()


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("seed", list(range(100, 112)) + list(range(300, 308)) + list(range(900, 912)))
def test_oracle_vs_reference_bullet_random(built, seed):
    """seeds >= 300: half of the shapes are z-cylinders (GT_Cylinder, bullet_collision_checker.cpp:153-158); seeds >= 900:
    crowded scenes, ~45 % spheres (GT_Sphere, :150-152): sphere vs box / hull / cylinder through GJK with the radius as
    margin, sphere vs sphere through btSphereSphereCollisionAlgorithm (no contact while the surfaces are apart), swept
    spheres, and the one cull of the reference that is not grown by d (swept sphere vs a compound's sphere child)."""
    rng = np.random.default_rng(seed)
    if seed >= 900:
        s, objT, robot = synth.random_scene(rng, spread=0.35, round_prob=0.9, spheres=True, n_static=8, n_robot=6)
    else:
        s, objT, robot = synth.random_scene(rng, spread=float(rng.choice([0.3, 0.6])), round_prob=0.5 if seed >= 300 else 0.0)
    wr, wo, we = B.ref_world(s), B.oracle_world(s), util.emu_world(s)
    T1 = objT[robot].copy()
    T1[:, 4:] += rng.normal(size=(len(robot), 3)) * 0.15
    for d in (0.0, 0.09):
        for w in (wr, wo, we):
            w.set_contact_distance(d)
        ref = wr.links_vs_all(objT, robot)
        util.assert_same_contacts(wo.links_vs_all(objT, robot), ref, "oracle disc")
        util.assert_same_contacts(we.links_vs_all(objT, robot), ref, "emu disc")
        refc = wr.cast_vs_all(objT, robot, objT[robot], T1)
        util.assert_same_contacts(wo.cast_vs_all(objT, robot, objT[robot], T1), refc, "oracle cast")
        util.assert_same_contacts(we.cast_vs_all(objT, robot, objT[robot], T1), refc, "emu cast")
    for w in (wr, wo, we):
        w.close()


def test_shape_hull_golden_fixture_bit_exact(built):
    """tohc_shape_hull (the btShapeHull reduction of a GT_TriMesh, bullet_collision_checker.cpp:174-203) against points
    recorded from the reference's own Bullet (tests/golden/make_shape_hull.py): values and order."""
    from trajopt_amd import engine
    z = np.load(os.path.join(ROOT, "tests", "golden", "shape_hull.npz"), allow_pickle=False)
    sizes = set()
    for k in range(int(z["n"])):
        v, t, h = z["v%d" % k], z["t%d" % k], z["h%d" % k]
        mine = engine.shape_hull(v, t)
        assert mine.shape == h.shape and np.array_equal(mine.view(np.uint32), h.view(np.uint32)), k
        if len(v) < 50:
            assert np.array_equal(mine, v)          # used verbatim (:189-201)
        else:
            assert len(mine) <= 42
        sizes.add(len(mine))
    assert 4 in sizes and max(sizes) >= 40           # flat meshes collapse to a quad, blobs keep ~42 points
    # bad index list
    lib = engine.load_library()
    v = np.zeros((3, 3), np.float32); t = np.array([[0, 1, 3]], np.int32)
    assert lib.tohc_shape_hull(v.ctypes.data_as(C.POINTER(C.c_float)), 3, t.ctypes.data_as(C.POINTER(C.c_int32)), 1, None, 0) == -1
    # a mesh without triangles is an error (the reference asserts indices.size() >= 3, bullet_collision_checker.cpp:162), whatever
    # its vertex count -- not HullLibrary's fall-back box for 60 unreferenced points
    big = np.random.default_rng(0).normal(size=(60, 3)).astype(np.float32)
    assert lib.tohc_shape_hull(big.ctypes.data_as(C.POINTER(C.c_float)), 60, t.ctypes.data_as(C.POINTER(C.c_int32)), 0, None, 0) == -1
    with pytest.raises(engine.EngineError):
        engine.shape_hull(big, np.zeros((0, 3), np.int32))


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
def test_shape_hull_vs_reference_bullet_random(built):
    from trajopt_amd import engine
    for seed in range(200):
        rng = np.random.default_rng(7100 + seed)
        v, t = synth.random_trimesh(rng, int(rng.choice([30, 50, 60, 200, 1000])), seed % 4)
        a, b = B.ref_shape_hull(v, t), engine.shape_hull(v, t)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), seed


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
def test_trimesh_objects_through_shape_hull_vs_reference_bullet(built):
    """A world whose objects are GT_TriMesh geometries of 60..800 vertices, reduced by tohc_shape_hull (Scene.add_trimesh):
    discrete and cast contacts of the oracle and of the host build of the device code equal those of the reference's
    Bullet on ITS btShapeHull points, bit for bit."""
    rng = np.random.default_rng(7300)
    s, sr = M.Scene(), M.Scene()
    objT, robot = [], []
    for i in range(10):
        v, t = synth.random_trimesh(rng, int(rng.choice([60, 200, 800])), int(rng.choice([0, 0, 1, 3])))
        v = ((v - v.mean(0)) * 0.5).astype(np.float32)
        grp = M.GROUP_ROBOT if i < 4 else M.GROUP_KINBODY
        s.add_object([(s.add_trimesh(v, t), M.IDENT7)], grp, name="mesh%d" % i)
        sr.add_object([(sr.add_hull(B.ref_shape_hull(v, t)), M.IDENT7)], grp, name="mesh%d" % i)
        if i < 4:
            robot.append(i)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        objT.append(np.concatenate([q, rng.uniform(-0.25, 0.25, 3)]))
    s.finalize(); sr.finalize()
    objT = np.array(objT)
    T1 = objT[robot].copy()
    T1[:, 4:] += rng.normal(size=(len(robot), 3)) * 0.1
    wr, wo, we = B.ref_world(sr), B.oracle_world(s), util.emu_world(s)
    for w in (wr, wo, we):
        w.set_contact_distance(0.09)
    ref = wr.links_vs_all(objT, robot)
    assert len(ref) > 5
    util.assert_same_contacts(wo.links_vs_all(objT, robot), ref, "oracle disc")
    util.assert_same_contacts(we.links_vs_all(objT, robot), ref, "emu disc")
    refc = wr.cast_vs_all(objT, robot, objT[robot], T1)
    util.assert_same_contacts(wo.cast_vs_all(objT, robot, objT[robot], T1), refc, "oracle cast")
    util.assert_same_contacts(we.cast_vs_all(objT, robot, objT[robot], T1), refc, "emu cast")
    for w in (wr, wo, we):
        w.close()


def test_epa_small_pools_then_full_pools_same_results(built, tmp_path):
    """k_tail_epa runs EPA on small pools first and repeats a run that leaves them on the reference's pools.  Here the
    host build of the device source is compiled with pools so small (6 support vertices, 10 faces) that most runs are
    repeated, on crowded scenes (deep penetrations): contact sets and values equal the oracle's (and the reference's
    Bullet where it is built) bit for bit -- the short cut is invisible."""
    import subprocess
    lib = str(tmp_path / "libemu_tiny.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas", "-DTOHC_EPA_SMALL_V=6",
                    "-DTOHC_EPA_SMALL_F=10", "-shared", "-o", lib, os.path.join(util.ROOT, "tests", "emu", "emu.cpp")], check=True)
    dll = C.CDLL(lib)
    dll.emu_epa_redone.restype = C.c_long
    dll.emu_counter.restype = C.c_long
    for seed in (101, 303, 905):
        rng = np.random.default_rng(seed)
        s, objT, robot = synth.random_scene(rng, spread=0.3, round_prob=0.5 if seed >= 300 else 0.0, spheres=seed >= 900, n_static=10, n_robot=6)
        worlds = [B.oracle_world(s), B.World(dll, "emu_", s)] + ([B.ref_world(s)] if B.ref_available() else [])
        T1 = objT[robot].copy()
        T1[:, 4:] += rng.normal(size=(len(robot), 3)) * 0.15
        for w in worlds:
            w.set_contact_distance(0.09)
        res = [(w.links_vs_all(objT, robot), w.cast_vs_all(objT, robot, objT[robot], T1)) for w in worlds]
        for r in res[1:]:
            util.assert_same_contacts(r[0], res[0][0], "disc")
            util.assert_same_contacts(r[1], res[0][1], "cast")
        assert (res[0][0]["dist"] < -0.01).any()
        for w in worlds:
            w.close()
    assert dll.emu_epa_redone() > 20, dll.emu_epa_redone()


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("seed", list(range(100, 106)) + list(range(300, 304)) + list(range(900, 906)))
def test_all_vs_all_oracle_vs_reference_bullet_random(built, seed):
    """AllVsAll on random worlds (boxes, hulls, cylinders, spheres, compounds on both sides): the oracle against the
    reference's manifold walk, twice in a row (the reference clears its manifolds after every call).  A fresh reference
    world per contact distance: its manifolds keep the breaking threshold they were created under."""
    rng = np.random.default_rng(seed)
    if seed >= 900:
        s, objT, robot = synth.random_scene(rng, spread=0.35, round_prob=0.9, spheres=True, n_static=8, n_robot=6)
    else:
        s, objT, robot = synth.random_scene(rng, spread=float(rng.choice([0.3, 0.6])), round_prob=0.5 if seed >= 300 else 0.0)
    n = 0
    for d in (0.0, 0.09):
        wr, wo = B.ref_world(s), B.oracle_world(s)
        wr.set_contact_distance(d); wo.set_contact_distance(d)
        for rep in range(2):
            a, b = wo.all_vs_all(objT), wr.all_vs_all(objT)
            util.assert_same_all_vs_all(a, b, "seed %d d %g" % (seed, d))
            n += len(a)
        wr.close(); wo.close()
    assert n > 10


@pytest.mark.parametrize("mk", WORLDS)
def test_reference_continuous_check_known_answer(built, mk):
    """collision-checker-unit.cpp:102-120: boxbot swept from (-1.9, 0) to (0, 1.9) past the unit box at the origin:
    one collision whose robot normal is (-1, 0, 0)."""
    s = M.Scene()
    b = s.add_box(.5, .5, .5)
    s.add_object([(b, M.IDENT7)], M.GROUP_KINBODY)
    bot = s.add_object([(b, M.IDENT7)], M.GROUP_ROBOT)
    s.finalize()
    w = mk(s)
    w.set_contact_distance(0.05)
    T = lambda x, y: (1, 0, 0, 0, x, y, 0)
    h = w.continuous_check([T(0, 0), T(-1.9, 0)], [bot], [[T(-1.9, 0)], [T(0, 1.9)]])
    assert len(h) == 1 and h[0]["obj"] == bot and h[0]["hitObj"] == 0
    assert np.allclose(h[0]["n"], (-1, 0, 0), atol=1e-4)
    assert h[0]["fraction"] == np.float32(0.4736842)      # what the reference's own Bullet returns (oracle/_ref)
    assert len(w.continuous_check([T(0, 0), T(-1.9, 0)], [bot], [[T(-1.9, 0)], [T(-1.9, 1.9)]])) == 0
    w.close()


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("seed", list(range(400, 412)) + list(range(920, 926)))
def test_continuous_check_oracle_vs_reference_bullet(built, seed):
    """ContinuousCheckTrajectory: the C restatement and the host build of the DEVICE code (emu) against the reference's
    own btCollisionWorld::convexSweepTest, bit for bit.  Where several objects share the smallest fraction (0: already
    touching at the start of the segment) the reference reports the first in DBVT order, which is history dependent;
    there its choice is checked pair by pair (ora_sweep_pair) instead."""
    rng = np.random.default_rng(seed)
    s, objT, robot = synth.random_scene(rng, spread=float(rng.choice([0.4, 0.8, 1.5])), round_prob=0.3 if seed % 2 else 0.0, spheres=False) if seed < 900 else \
        synth.random_scene(rng, spread=float(rng.choice([0.4, 0.8])), round_prob=0.8, spheres=True)   # seeds >= 900: ~40 % spheres
    nst = 4
    traj = np.stack([np.array([synth.random_T7(rng, -0.8, 0.8) for _ in robot]) for _ in range(nst)])
    wr, wo, we = B.ref_world(s), B.oracle_world(s), util.emu_world(s)
    n = 0
    for d in (0.0, 0.05):
        for w in (wr, wo, we):
            w.set_contact_distance(d)
        a, o, e = (w.continuous_check(objT, robot, traj) for w in (wr, wo, we))
        assert o.tobytes() == e.tobytes()
        key = lambda h: [(int(x["obj"]), int(x["child"]), int(x["seg"]), float(x["fraction"])) for x in h]
        assert key(a) == key(o)
        for x, y in zip(a, o):
            if x.tobytes() == y.tobytes():
                continue
            assert x["fraction"] == 0 and (x["hitObj"], x["hitChild"]) > (y["hitObj"], y["hitChild"])   # a tie, oracle = lowest
            qi = list(robot).index(int(x["obj"]))
            p = wo.sweep_pair(x["obj"], x["child"], traj[x["seg"], qi], traj[x["seg"] + 1, qi], x["hitObj"], x["hitChild"], objT[x["hitObj"]])
            assert p is not None and p["fraction"] == 0 and np.array_equal(p["n"], x["n"]) and np.array_equal(p["pt"], x["pt"])
        n += len(a)
    assert n > 0 or seed in (400,)      # (a sparse draw: nothing is hit)
    for w in (wr, wo, we):
        w.close()


def test_portable_trig_is_correctly_rounded_and_shared(built):
    """The device cannot call glibc's sinf / cosf / acosf; it runs portable fp64-series routines (tohc_narrow.h).  Here:
    they return the correctly rounded float over the ranges the advancement uses, and the oracle / the host build of the
    device code, switched to them, still agree bit for bit (what makes the GPU test bit-exact)."""
    import ctypes as C
    lib = C.CDLL(os.path.join(util.ROOT, "tests", "emu", "libemu.so"))
    lib.emu_pt_trig.restype = C.c_float
    lib.emu_pt_trig.argtypes = [C.c_int, C.c_float]
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-0.4, 0.4, 4000), rng.uniform(-1e-3, 1e-3, 500), [0.0, 0.39269908, -0.39269908]]).astype(np.float32)
    for x in xs:
        assert np.float32(lib.emu_pt_trig(0, float(x))) == np.float32(np.sin(np.float64(x)))
        assert np.float32(lib.emu_pt_trig(1, float(x))) == np.float32(np.cos(np.float64(x)))
    ws = np.concatenate([rng.uniform(-1, 1, 4000), 1 - rng.uniform(0, 1e-5, 1000), [1.0, -1.0, 0.0, 0.5, -0.5, 0.99999994]]).astype(np.float32)
    for w in ws:
        assert np.float32(lib.emu_pt_trig(2, float(w))) == np.float32(np.arccos(np.float64(w))), w
    rng = np.random.default_rng(431)
    s, objT, robot = synth.random_scene(rng, spread=0.8, round_prob=0.3)
    traj = np.stack([np.array([synth.random_T7(rng, -0.8, 0.8) for _ in robot]) for _ in range(4)])
    wo, we = B.oracle_world(s), util.emu_world(s)
    try:
        for w in (wo, we):
            w.set_trig_mode(1)
            w.set_contact_distance(0.05)
        o, e = wo.continuous_check(objT, robot, traj), we.continuous_check(objT, robot, traj)
        assert len(o) > 0 and o.tobytes() == e.tobytes()
    finally:
        for w in (wo, we):
            w.set_trig_mode(0)
            w.close()


def test_continuous_check_libm_vs_portable_trig_bound(built):
    """How far the device's ContinuousCheckTrajectory can be from the reference's because it cannot call glibc's float
    trig: the oracle in libm mode (bit-exact against the reference's convexSweepTest, see above) against the oracle in
    portable-trig mode (bit-exact against the GPU, tests/test_gpu_parity.py), on 120 random sweeps WITH rotation.
    Measured here: 1525 common hits, 75 (4.9 %) with a different fraction, the largest difference 0.069 of a segment; one
    (segment, link geometry) hit in one mode only, two hits on a different object.  The conservative advancement is
    discontinuous in the last bit of sin / cos / acos; sweeps without rotation never call them and are
    bit-identical.  The bounds asserted are those measurements with headroom."""
    tot = moved = setdiff = otherobj = 0
    maxd = 0.0
    for seed in range(700, 760):
        rng = np.random.default_rng(seed)
        s, objT, robot = synth.random_scene(rng, spread=float(rng.choice([0.4, 0.8, 1.5])), round_prob=0.3 if seed % 2 else 0.0)
        traj = np.stack([np.array([synth.random_T7(rng, -0.8, 0.8) for _ in robot]) for _ in range(5)])
        w = B.oracle_world(s)
        try:
            for d in (0.0, 0.05):
                w.set_contact_distance(d)
                w.set_trig_mode(0)
                a = w.continuous_check(objT, robot, traj)
                w.set_trig_mode(1)
                b = w.continuous_check(objT, robot, traj)
                ka = {(int(x["seg"]), int(x["obj"]), int(x["child"])): x for x in a}
                kb = {(int(x["seg"]), int(x["obj"]), int(x["child"])): x for x in b}
                setdiff += len(set(ka) ^ set(kb))
                for k in set(ka) & set(kb):
                    tot += 1
                    df = abs(float(ka[k]["fraction"]) - float(kb[k]["fraction"]))
                    moved += df > 0
                    maxd = max(maxd, df)
                    otherobj += (int(ka[k]["hitObj"]), int(ka[k]["hitChild"])) != (int(kb[k]["hitObj"]), int(kb[k]["hitChild"]))
        finally:
            w.set_trig_mode(0)
            w.close()
    assert tot > 1000
    assert setdiff <= 0.005 * tot and otherobj <= 0.005 * tot      # <= 0.5 % of the hits appear / disappear / change object
    assert moved <= 0.08 * tot and maxd <= 0.1                     # <= 8 % of the fractions move, by at most a tenth of a segment
    # pure translations: no trig, bit-identical
    rng = np.random.default_rng(761)
    s, objT, robot = synth.random_scene(rng, spread=0.8, round_prob=0.3)
    traj = np.tile(objT[robot][None], (5, 1, 1))
    traj[:, :, 4:] += np.cumsum(rng.normal(size=(5, len(robot), 3)) * 0.3, axis=0)
    w = B.oracle_world(s)
    try:
        w.set_contact_distance(0.05)
        w.set_trig_mode(0)
        a = w.continuous_check(objT, robot, traj)
        w.set_trig_mode(1)
        b = w.continuous_check(objT, robot, traj)
        assert len(a) > 0 and a.tobytes() == b.tobytes()
    finally:
        w.set_trig_mode(0)
        w.close()


def test_allowed_matrix_and_filter_mask(built):
    rng = np.random.default_rng(7)
    s, objT, robot = synth.random_scene(rng, spread=0.3)
    w = B.oracle_world(s)
    w.set_contact_distance(0.2)
    full = w.links_vs_all(objT, robot)
    assert len(full) > 0
    only_env = w.links_vs_all(objT, robot, mask=M.GROUP_KINBODY)
    groups = np.array(s.obj_group)
    assert all(groups[c["objA"]] == M.GROUP_KINBODY or groups[c["objB"]] == M.GROUP_KINBODY for c in only_env)
    assert len(only_env) < len(full)
    m = np.ones((s.n_objects, s.n_objects), np.uint8)
    a, b = int(full[0]["objA"]), int(full[0]["objB"])
    m[a, b] = m[b, a] = 0
    w.set_allowed(m)
    cut = w.links_vs_all(objT, robot)
    assert all({int(c["objA"]), int(c["objB"])} != {a, b} for c in cut) and len(cut) < len(full)
    w.close()


def test_empty_query_and_far_apart(built):
    s, o0, o1 = two_boxes(M.GROUP_KINBODY)
    w = B.oracle_world(s)
    assert len(w.links_vs_all([T(0), T(50)], [o0])) == 0
    assert len(w.links_vs_all([T(0), T(.5)], [])) == 0
    w.close()


# ---- FK / Jacobian ------------------------------------------------------------------------------------------
def _quat_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("active", ["rightarm", "fullbody"])
def test_jacobian_matches_finite_differences(built, active):
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active=active)
    s.finalize(); rb.finalize()
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, size=rb.K)
    _, links = rb.affected_links()
    for link in links[::4]:
        pl = rng.normal(size=3) * 0.1

        def world_pt(qq):
            Tl = B.oracle_fk(rb, qq)[link]
            return _quat_R(Tl[:4]) @ pl + Tl[4:]

        J = B.oracle_position_jacobian(rb, q, link, world_pt(q))
        Jfd = np.stack([(world_pt(q + h) - world_pt(q - h)) / 2e-6 for h in np.eye(rb.K) * 1e-6], axis=1)
        assert np.abs(J - Jfd).max() < 1e-7


def test_fk_planar_three_links_closed_form(built):
    # data/three_links.robot.xml: planar 3R arm, link lengths 0.16, 0.16
    s = M.Scene()
    rb = synth.three_links(s)
    s.finalize(); rb.finalize()
    q = np.array([0.3, -0.7, 1.1])
    Tl = B.oracle_fk(rb, q)
    a1, a2 = q[0], q[0] + q[1]
    assert np.allclose(Tl[2][4:], [0.16 * np.cos(a1), 0.16 * np.sin(a1), 0], atol=1e-14)
    assert np.allclose(Tl[3][4:], [0.16 * np.cos(a1) + 0.16 * np.cos(a2), 0.16 * np.sin(a1) + 0.16 * np.sin(a2), 0], atol=1e-14)
    ang = q.sum()
    assert np.allclose(Tl[3][:4], [np.cos(ang / 2), 0, 0, np.sin(ang / 2)], atol=1e-14)
    # free rigid body known answer (src/sandbox/incremental_rb.hpp:45-54): J = [I | skew] -> for the planar base
    s2 = M.Scene()
    bot = synth.boxbot(s2)
    bot.set_active([], M.DOF_X | M.DOF_Y | M.DOF_ROTAXIS)
    s2.finalize(); bot.finalize()
    J = B.oracle_position_jacobian(bot, np.array([.2, .3, .4]), 0, np.array([1.0, 2.0, 3.0]))
    assert np.allclose(J, [[1, 0, -(2.0 - .3)], [0, 1, (1.0 - .2)], [0, 0, 0]])


def test_linearization_is_first_order_accurate(built):
    """const + grad.x reproduces the signed distance at x and predicts it nearby (collision_terms.cpp:37-62)."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    ids, boxT = synth.add_boxes(s, [((.85, .55, .01), synth.T7((1.1, 0, .635)))])
    s.finalize(); rb.finalize()
    static_T = np.tile(np.array(M.IDENT7), (s.n_objects, 1))
    static_T[ids] = boxT
    w = B.oracle_world(s)
    w.set_contact_distance(0.2)
    q = np.array([-0.4, 0.6, -0.5, -1.2, -1.0, -1.0, 1.0])
    rec, g = w.linearize(rb, static_T, q[None], mask=M.GROUP_KINBODY)
    assert len(rec) > 0
    assert np.allclose(rec["const0"] + g[:, 0] @ q, rec["c"]["dist"], atol=1e-12)
    dq = np.full(7, 1e-4)
    rec2, _ = w.linearize(rb, static_T, (q + dq)[None], mask=M.GROUP_KINBODY)
    k1 = {tuple(int(r["c"][k]) for k in util.KEYS): i for i, r in enumerate(rec)}
    hits = 0
    for r in rec2:
        i = k1.get(tuple(int(r["c"][k]) for k in util.KEYS))
        if i is not None and rec[i]["c"]["dist"] > 1e-3 and r["c"]["dist"] > 1e-3:
            pred = rec[i]["const0"] + g[i, 0] @ (q + dq)
            assert abs(pred - r["c"]["dist"]) < 5e-6
            hits += 1
    assert hits > 0
    w.close()


def test_sanitizer_run_of_oracle_and_device_source(built):
    """the C oracle and the host build of the DEVICE narrowphase source under AddressSanitizer +
    UndefinedBehaviorSanitizer (`make -C oracle san`: a standalone driver over random scenes with boxes, 4..64-vertex
    hulls, spheres, cylinders and compounds -- discrete, cast, sweeps, rays, FK + linearization), which also compares the
    two bit for bit.  Any sanitizer report aborts the driver."""
    import subprocess
    p = subprocess.run(["make", "-C", os.path.join(util.ROOT, "oracle"), "san"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "san_driver: OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr


# ---- C-ABI surface ------------------------------------------------------------------------------------------
def test_abi_library_exports_every_declared_symbol(built):
    from trajopt_amd import engine
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "trajopt_hipcc.h")).read()
    declared = set(re.findall(r"\b(tohc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(engine.SYMBOLS), declared ^ set(engine.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_engine_fails_loudly_without_gpu(built):
    import torch
    from trajopt_amd import engine
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.EngineError):
        engine.Engine(0)


def test_cpp_factory_switch_refuses_other_checkers(built):
    """CreateCollisionChecker (the reference's one factory, bullet_collision_checker.cpp:1036-1039) with the switch
    TRAJOPT_COLLISION_CHECKER: asking for `bullet` (not part of this build) or an unknown name throws -- no GPU needed."""
    import subprocess
    drv = os.path.join(ROOT, "tests", "cpp", "host_unit")
    p = subprocess.run([drv, "factory_switch"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "PASSED" in p.stdout, p.stdout + p.stderr
