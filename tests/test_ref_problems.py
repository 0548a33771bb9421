"""BASELINE.json configs[0] / configs[1] on their OWN initial trajectories: data/arm_around_table.json:33-42 (T = 10) and
data/arm_around_table_continuous.json:36-41 (T = 6), committed as numbers in tests/golden/ref_problems.json (made by
tests/golden/make_problems.py), in the world of data/table.xml, d = dist_pen 0.025 + 0.04 (problem_description.cpp:548-549).
Both files are CAST problems as shipped ("continuous" defaults to true, problem_description.cpp:493); BASELINE.json words
configs[0] as discrete, so each trajectory is run both ways.

CPU: the oracle against the reference's Bullet (oracle/_ref), bit for bit.  GPU: the engine against the oracle.
(The robot is the build's PR2-like right arm: OpenRAVE's PR2 model is not in the image.)
"""
import numpy as np
import pytest

from oracle import bindings as B
from tests import util
from trajopt_amd import synth

PROBLEMS = ["arm_around_table", "arm_around_table_continuous"]


@pytest.mark.parametrize("name", PROBLEMS)
def test_fixture_holds_the_reference_numbers(name):
    s, rb, static_T, d, traj, p = synth.config_ref_problem(name)
    assert traj.shape == (p["n_steps"], 7) == ({"arm_around_table": 10, "arm_around_table_continuous": 6}[name], 7)
    assert p["continuous"] is True and d == 0.065 and p["coeffs"] == [20]
    assert np.array_equal(traj[-1], p["joint_target"])              # the straight line ends at the joint constraint's values
    assert s.arrays["obj_compound"][-1] == 1 and s.arrays["obj_cnt"][-1] == 1   # table: one geometry off the link's origin


@pytest.mark.skipif(not B.ref_available(), reason="oracle/_ref/libref.so is only built where /root/reference exists")
@pytest.mark.parametrize("name", PROBLEMS)
def test_oracle_vs_reference_bullet_on_the_json_trajectories(built, name):
    s, rb, static_T, d, traj, _ = synth.config_ref_problem(name)
    wo, wr = B.oracle_world(s), B.ref_world(s)
    for w in (wo, wr):
        w.set_contact_distance(d)
    objs, links = rb.affected_links()
    link_obj = np.array(rb.link_obj)
    have = link_obj >= 0
    total = 0
    for t in range(len(traj)):
        fk0 = B.oracle_fk(rb, traj[t])
        P = static_T.copy()
        P[link_obj[have]] = fk0[have]
        a, b = wo.links_vs_all(P, objs), wr.links_vs_all(P, objs)
        util.assert_same_contacts(a, b, "%s discrete t=%d" % (name, t))
        total += len(a)
        if t + 1 < len(traj):
            fk1 = B.oracle_fk(rb, traj[t + 1])
            ac, bc = wo.cast_vs_all(P, objs, fk0[links], fk1[links]), wr.cast_vs_all(P, objs, fk0[links], fk1[links])
            util.assert_same_contacts(ac, bc, "%s cast t=%d" % (name, t))
            total += len(ac)
    assert total > 0
    wo.close(); wr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROBLEMS)
def test_engine_vs_oracle_on_the_json_trajectories(built, name):
    """One SQP linearization of the problem at its init trajectory, as the reference's first convexify sees it: the cast cost
    (as shipped) and the discrete cost (BASELINE.json's wording), contacts bit for bit, gradient rows to 1e-5; then the
    same slab through tohc_enqueue_linearization."""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d, traj, _ = synth.config_ref_problem(name)
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb)                       # + IgnoreZeroStateSelfCollisions, like the reference's AddKinBody path
    cc.SetContactDistance(d)
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    T = len(traj)
    n = {}
    for cast in (True, False):
        rec, g = cc.eng.cast_vs_all_batch(traj, 1, T, 1) if cast else cc.eng.links_vs_all_batch(traj, 1, T)
        orec, og = w.linearize(rb, static_T, traj, cast=cast)
        util.assert_same_linearization(rec, g, orec, og, "%s cast=%s" % (name, cast))
        n[cast] = (rec, g)
    assert len(n[True][0]) + len(n[False][0]) > 0
    cc.eng.stage_dofs(traj, 1, T)
    cc.eng.enqueue_linearization(-1, 1, True)
    both, gboth = cc.eng.fetch()
    disc = both["cctype"] == 0
    assert np.array_equal(both[disc], n[False][0]) and np.array_equal(both[~disc], n[True][0])
    assert np.array_equal(gboth[disc], n[False][1]) and np.array_equal(gboth[~disc], n[True][1])
    w.close()
    cc.eng.close()
