"""The C++ host layer (trajopt_amd/host: HIPCollisionChecker, CollisionPrefetchCost over the C-ABI) on the GPU.

tests/cpp/host_unit is built by __graft_entry__.build().  Its first two commands ARE the reference's unit tests
(collision-checker-unit.cpp:29-153); `linearize` asks the checker what the reference's collision terms ask it
(collision_terms.cpp: LinksVsAll per timestep, CastVsAll per pair) on a world written here, call by call and behind a
CollisionPrefetchCost, and dumps every contact and hinge row, which are compared bit for bit with the Python host mirror
(same C-ABI underneath) and, to the north-star tolerances, with the CPU oracle; `fk_guard` / `grab_rebind` test the
checker's FK guard and its re-binding after Grab / Release.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import bindings as B
from tests import util
from trajopt_amd import collision_checker as CC, model as M, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "host_unit")

pytestmark = pytest.mark.gpu


def _run(*args):
    assert os.path.exists(DRIVER), "tests/cpp/host_unit is not built: run __graft_entry__.build()"
    p = subprocess.run([DRIVER] + list(args), capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, "host_unit %s failed (rc %d):\n%s\n%s" % (args[0], p.returncode, p.stdout[-3000:], p.stderr[-2000:])
    return p.stdout


@pytest.mark.parametrize("cmd", ["box_distance", "continuous_boxes", "factory_switch", "factory_hip", "fk_guard", "grab_rebind"])
def test_reference_unit_tests_through_the_cpp_host(built, cmd):
    assert "PASSED" in _run(cmd)


def _world(cfg):
    if cfg == "arm_table":      # configs[0] / [1]: right arm (K = 7) over data/table.xml + a box in reach, 10 timesteps
        s = M.Scene()
        rb = synth.pr2_like(s, seed=0, active="rightarm")
        ids, boxT = synth.add_boxes(s, [synth.TABLE_BOX, ((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))])
        s.finalize(); rb.finalize()
        static_T = synth.static_poses(s, ids, boxT)
        rng = np.random.default_rng(11)
        q0 = rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, size=7)
        traj = synth.straight_line(q0, q0 + rng.uniform(-0.9, 0.9, size=7), 10)
        return s, rb, static_T, 0.065, traj, 0.025
    if cfg == "arm_grab_spheres":   # the arm holding a box (robot->Grab, configuration_space.cpp:120-131) among spheres and a cylinder
        s = M.Scene()
        rb = synth.pr2_like(s, seed=0, active="rightarm")
        rng = np.random.default_rng(12)
        held = s.add_object([(s.add_box(0.03, 0.03, 0.12), M.IDENT7)], M.GROUP_KINBODY, name="held")
        ids, Ts = [], []
        for i in range(6):
            ids.append(s.add_object([(s.add_sphere(rng.uniform(0.05, 0.15)), M.IDENT7)], M.GROUP_KINBODY))
            Ts.append(synth.T7(rng.uniform([0.3, -0.8, 0.4], [0.9, 0.2, 1.2])))
        ids.append(s.add_object([(s.add_cylinder(0.06, 0.3), M.IDENT7)], M.GROUP_KINBODY))
        Ts.append(synth.T7((0.6, -0.4, 0.7)))
        rb.grab(rb.link_name.index("r_gripper_palm_link"), synth.T7((0.12, 0, 0), (0, 1, 0), 0.3), held)
        s.finalize(); rb.finalize()
        static_T = synth.static_poses(s, ids, np.array(Ts))
        q0 = rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, size=7)
        traj = synth.straight_line(q0, q0 + rng.uniform(-0.9, 0.9, size=7), 8)
        return s, rb, static_T, 0.09, traj, 0.05
    if cfg == "arm_meshes":     # GT_TriMesh obstacles of 60..800 vertices: the checker's btShapeHull reduction (tohc_shape_hull, :174-203)
        s = M.Scene()
        rb = synth.pr2_like(s, seed=0, active="rightarm")
        rng = np.random.default_rng(13)
        ids, Ts = [], []
        for i in range(8):
            v, t = synth.random_trimesh(rng, int(rng.choice([40, 60, 200, 800])), int(rng.choice([0, 1, 3])))
            v = ((v - v.mean(0)) * 0.4).astype(np.float32)
            ids.append(s.add_object([(s.add_trimesh(v, t), M.IDENT7)], M.GROUP_KINBODY))
            Ts.append(synth.T7(rng.uniform([0.3, -0.8, 0.4], [0.9, 0.2, 1.2]), rng.normal(size=3), rng.uniform(0, 3)))
        s.finalize(); rb.finalize()
        static_T = synth.static_poses(s, ids, np.array(Ts))
        q0 = rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, size=7)
        traj = synth.straight_line(q0, q0 + rng.uniform(-0.9, 0.9, size=7), 8)
        return s, rb, static_T, 0.09, traj, 0.05
    s, rb, static_T, d = synth.config_fullbody_soup(n_hulls=256)    # configs[3] with a smaller soup, 12 timesteps
    traj = synth.fullbody_trajectory(rb, np.random.default_rng(1000), 12)
    return s, rb, static_T, d, traj, 0.05


KEY = ("t", "idA", "childA", "idB", "childB")


@pytest.mark.parametrize("cfg", ["arm_table", "fullbody_soup256", "arm_grab_spheres", "arm_meshes"])
def test_cpp_collision_terms_vs_python_host_and_oracle(built, tmp_path, cfg):
    s, rb, static_T, d, traj, dist_pen = _world(cfg)
    coeff = 20.0
    fin, fout = str(tmp_path / "world.txt"), str(tmp_path / "dump.txt")
    util.write_flat_world(fin, s, rb, static_T, traj, d, dist_pen, coeff)
    assert "PASSED" in _run("linearize", fin, fout)
    meta, costs, sweep = util.read_host_unit_dump(fout)
    T, K = traj.shape
    assert (meta["T"], meta["K"]) == (T, K)
    # call by call: every term queries in the evaluate and in the convexify pass (+ the FK guard of the robot upload); behind
    # the CollisionPrefetchCost: ONE engine call, every query answered from its slab
    assert meta["calls_prefetched"] == 1 and meta["prefetch_hits"] == 2 * (2 * T - 1) and meta["calls_per_timestep"] == 2 * (2 * T - 1) + 1

    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb)
    cc.SetContactDistance(d)
    lin = CC.TrajectoryLinearizer(cc)
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    total = 0
    for cast in (False, True):
        rec, g = lin.linearize(traj, cast=cast)
        orec, og = w.linearize(rb, static_T, traj, cast=cast)
        util.assert_same_linearization(rec, g, orec, og, "%s cast=%s" % (cfg, cast))          # GPU vs oracle
        const, c0, c1 = CC.aff_exprs(rec, g, cast)
        mine = [c for c in costs if c["kind"] == ("cast" if cast else "disc")]
        assert [c["t"] for c in mine] == list(range(T - 1 if cast else T))
        rows = sorted(((c["t"], x["idA"], x["childA"], x["idB"], x["childB"]), x) for c in mine for x in c["contacts"])
        assert [r[0] for r in rows] == list(zip(*[rec[k].tolist() for k in KEY])), "C++ host and Python host see different contact sets"
        for i, (_, x) in enumerate(rows):                                                    # C++ host vs Python host: bit for bit
            assert x["dist"] == rec["dist"][i] and x["cctype"] == rec["cctype"][i] and x["time"] == rec["time"][i]
            for f in ("n", "ptA", "ptB", "ptB1"):
                assert x[f] == rec[f][i].tolist(), f
            t = int(rec["t"][i])
            assert x["const"] == dist_pen - const[i]
            if cast:
                assert x["vars"] == list(range(t * K, (t + 2) * K))
                assert x["coeffs"] == (-c0[i]).tolist() + (-c1[i]).tolist()
            else:
                assert x["vars"] == list(range(t * K, (t + 1) * K)) and x["coeffs"] == (-c0[i]).tolist()
        for c in mine:                                                                       # CollisionCost::value
            sel = rec["t"] == c["t"]
            assert c["n"] == int(sel.sum())
            assert c["value"] == pytest.approx(float(np.maximum(0.0, dist_pen - rec["dist"][sel]).sum() * coeff), rel=1e-14, abs=1e-15)
        total += len(rec)
    assert total > 0
    hits = cc.eng.continuous_check_trajectory(traj, 1, T)                                     # ContinuousCheckTrajectory
    a = sorted((h["idA"], h["childA"], h["idB"], h["childB"], float(h["toi"])) for h in sweep)
    b = sorted((int(r["idA"]), int(r["childA"]), int(r["idB"]), int(r["childB"]), float(np.float32(r["t"]) + np.float32(r["time"]))) for r in hits)
    assert a == b
    w.close()
    cc.eng.close()
