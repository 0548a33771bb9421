"""BASELINE.json configs[2]: python_examples/benchmark.py-shaped right-arm problems in the reference's own box worlds
(data/{tabletop,bookshelves,kitchen_counter,kitchen}.env.xml, committed as numbers in tests/golden/ref_scenes.json),
discrete + cast, d = dist_pen 0.05 + 0.04 (benchmark.py:71-104, problem_description.cpp:549).

CPU: the oracle against the reference's Bullet (oracle/_ref), bit for bit.  GPU: the engine against the oracle.
"""
import numpy as np
import pytest

from oracle import bindings as B
from tests import util
from trajopt_amd import model as M, synth

SCENES = ["tabletop", "bookshelves", "kitchen_counter", "kitchen"]
D = 0.09


def world(name):
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    util.add_ref_scene(s, name)
    s.finalize(); rb.finalize()
    return s, rb, synth.static_poses(s)


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("name", SCENES)
def test_oracle_vs_reference_bullet_in_reference_scenes(built, name):
    s, rb, static_T = world(name)
    wo, wr = B.oracle_world(s), B.ref_world(s)
    for w in (wo, wr):
        w.set_contact_distance(D)
    objs, links = rb.affected_links()
    link_obj = np.array(rb.link_obj)
    have = link_obj >= 0
    trajs = util.benchmark_trajectories(rb, np.random.default_rng(1), 3)
    total = 0
    for traj in trajs:
        for t in range(len(traj) - 1):
            fk0, fk1 = B.oracle_fk(rb, traj[t]), B.oracle_fk(rb, traj[t + 1])
            P = static_T.copy()
            P[link_obj[have]] = fk0[have]
            a, b = wo.links_vs_all(P, objs), wr.links_vs_all(P, objs)
            util.assert_same_contacts(a, b, "%s discrete t=%d" % (name, t))
            ac, bc = wo.cast_vs_all(P, objs, fk0[links], fk1[links]), wr.cast_vs_all(P, objs, fk0[links], fk1[links])
            util.assert_same_contacts(ac, bc, "%s cast t=%d" % (name, t))
            total += len(a) + len(ac)
    assert total > 0
    wo.close(); wr.close()


def _pose_of(rb, static_T, q):
    link_obj = np.array(rb.link_obj)
    have = link_obj >= 0
    P = static_T.copy()
    P[link_obj[have]] = B.oracle_fk(rb, q)[have]
    return P


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("name", SCENES)
def test_all_vs_all_oracle_vs_reference_bullet(built, name):
    """AllVsAll (bullet_collision_checker.cpp:463-498): the oracle's restatement against the reference's
    performDiscreteCollisionDetection + manifold walk (ref_driver.cpp: ref_all_vs_all), robot self pairs included:
    contact set, manifold order of the two objects, distance / points / normal bit for bit, weight 1 / numContacts = 1
    throughout, contacts beyond d (up to 2.001 d) present."""
    s, rb, static_T = world(name)
    trajs = util.benchmark_trajectories(rb, np.random.default_rng(1), 2)
    wo, wr = B.oracle_world(s), B.ref_world(s)
    for w in (wo, wr):
        w.set_contact_distance(D)
    # zero-state self collisions off, as trajopt does at set-up (collision_checker.cpp:41-56): asymmetric entries included
    allowed = np.ones((s.n_objects, s.n_objects), np.uint8)
    z = wo.links_vs_all(_pose_of(rb, static_T, np.zeros(rb.K)), list(range(s.n_objects)), mask=M.GROUP_ROBOT)
    allowed[z["objA"], z["objB"]] = 0
    for use_allowed in (False, True):
        if use_allowed:
            wo.set_allowed(allowed); wr.set_allowed(allowed)
        for q in (trajs[0, 0], trajs[1, 5]):
            P = _pose_of(rb, static_T, q)
            a, b = wo.all_vs_all(P), wr.all_vs_all(P)
            util.assert_same_all_vs_all(a, b, name)
            assert len(a) > 50 and (a["dist"] > D).any() and (a["dist"] <= np.float32(2.001 * np.float32(D))).all()
            pairs = list(zip(a["objA"].tolist(), a["childA"].tolist(), a["objB"].tolist(), a["childB"].tolist()))
            assert len(pairs) == len(set(pairs))
    wo.close(); wr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_engine_vs_oracle_in_reference_scenes(built, name):
    from trajopt_amd import collision_checker as CC
    s, rb, static_T = world(name)
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb)
    cc.SetContactDistance(D)
    w = B.oracle_world(s)
    w.set_contact_distance(D)
    w.set_allowed(cc.allowed)
    trajs = util.benchmark_trajectories(rb, np.random.default_rng(1), 8)
    Bn, T, K = trajs.shape
    total = 0
    for cast in (False, True):
        rec, g = (cc.eng.cast_vs_all_batch(trajs, Bn, T, 1) if cast else cc.eng.links_vs_all_batch(trajs, Bn, T))
        for b in range(Bn):
            orec, og = w.linearize(rb, static_T, trajs[b], cast=cast)
            sel = rec["problem"] == b
            util.assert_same_linearization(rec[sel], g[sel], orec, og, "%s cast=%s b=%d" % (name, cast, b))
            total += len(orec)
    assert total > 0
    # AllVsAll (bullet_collision_checker.cpp:463-498) in two poses: the engine against the oracle (which equals the reference's
    # manifold walk, test_all_vs_all_oracle_vs_reference_bullet): contact set with child numbers, values bit for bit, weight 1
    for q in (trajs[0, 0], trajs[1, 5]):
        cc.SetDOFValues(q)
        allc = []
        cc.AllVsAll(allc)
        assert all(c.weight == 1.0 for c in allc)
        rec = cc.eng.all_vs_all_poses(cc._current_poses())
        assert len(rec) == len(allc) > 50
        util.assert_same_contacts(util.records_to_contacts(rec), w.all_vs_all(cc._current_poses()), "%s AllVsAll" % name)
    w.close()
    cc.eng.close()
