"""The collision path inside a trust-region SQP (tests/sqp_harness.py = sco's loop restated as test infrastructure).

CPU: the reference's cast-cost-unit test (src/trajopt/test/cast-cost-unit.cpp:37-81, fixture data/box_cast_test.json)
driven by the oracle.  GPU: the same problems driven by the HIP engine; north-star gate "the final SQP trajectories
match to within the same tolerance" -- HIP-driven and oracle-driven runs are compared step for step.
"""
import numpy as np
import pytest

from oracle import bindings as B
from tests import sqp_harness as S
from trajopt_amd import model as M, synth

# data/box_cast_test.json: n_steps 3, start_fixed, joint_vel coeffs [1], collision {continuous, coeffs [10], dist_pen [0.2]},
# joint constraint vals [1.9, 3.8], init given_traj
BOX_CAST = dict(traj=np.array([[-1.9, 0.0], [0.0, 1.9], [1.9, 3.8]]), vel=[1.0], coeff=10.0, dist_pen=0.2)


def box_cast_world():
    s = M.Scene()
    box = s.add_object([(s.add_box(.5, .5, .5), M.IDENT7)], M.GROUP_KINBODY, name="box")     # data/box.xml
    rb = synth.boxbot(s)                                                                      # data/boxbot.xml
    s.finalize(); rb.finalize()
    return s, rb, synth.static_poses(s), BOX_CAST["dist_pen"] + 0.04                          # problem_description.cpp:549


def arm_world():
    """configs[1]-shaped: PR2-like right arm, 10 timesteps, table + a box the straight line runs into; discrete + cast."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    ids, boxT = synth.add_boxes(s, [synth.TABLE_BOX, ((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))])
    s.finalize(); rb.finalize()
    rng = np.random.default_rng(11)
    q0 = rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, size=7)
    traj = synth.straight_line(q0, q0 + rng.uniform(-0.9, 0.9, size=7), 10)
    return s, rb, synth.static_poses(s, ids, boxT), 0.025 + 0.04, traj


def oracle_sweep_hits(w, s, rb, static_T, traj):
    objs, links = rb.affected_links()
    poses = np.stack([B.oracle_fk(rb, q)[links] for q in traj])
    return w.continuous_check(static_T, objs, poses)


def test_qp_solver_known_answers():
    # min (x-3)^2 s.t. x <= 1  -> 1 ;  hinge problem: min x^2/2 + 10*max(0, 1-x) -> x = 1
    assert S.solve_qp(np.array([[2.0]]), np.array([-6.0]), np.eye(1), np.array([-np.inf]), np.array([1.0]))[0] == pytest.approx(1.0, abs=1e-8)
    P = np.diag([1.0, 0.0]); q = np.array([0.0, 10.0])
    A = np.array([[0.0, 1.0], [-1.0, -1.0]])
    z = S.solve_qp(P, q, A, np.array([0.0, -np.inf]), np.array([np.inf, -1.0]))
    assert z == pytest.approx([1.0, 0.0], abs=1e-7)


def test_cast_cost_unit_with_the_oracle(built):
    """cast-cost-unit.cpp:37-81: the straight line drags boxbot through the box; after SQP the trajectory has 0
    continuous collisions (and the box is cleared by about dist_pen)."""
    s, rb, static_T, d = box_cast_world()
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    assert len(oracle_sweep_hits(w, s, rb, static_T, BOX_CAST["traj"])) == 1            # collision-checker-unit.cpp:116
    prob = S.TrajProblem(BOX_CAST["traj"], BOX_CAST["vel"], BOX_CAST["dist_pen"], BOX_CAST["coeff"], S.oracle_linearizer(w, rb, static_T),
                         discrete=False, cast=True)
    log = []
    x, cost, status = S.optimize(prob, log=log)
    assert status == "converged" and cost < log[0][2]
    assert np.array_equal(x[0], BOX_CAST["traj"][0]) and np.array_equal(x[-1], BOX_CAST["traj"][-1])
    assert len(oracle_sweep_hits(w, s, rb, static_T, x)) == 0                            # ASSERT_EQ(collisions.size(), 0)
    _, dist, *_ = prob.lin(x, True)
    assert dist.min(initial=1.0) > 0.1
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box_cast", "arm_table"])
def test_sqp_trajectory_hip_vs_oracle(built, name):
    from trajopt_amd import collision_checker as CC
    if name == "box_cast":
        s, rb, static_T, d = box_cast_world()
        traj0, kw = BOX_CAST["traj"], dict(vel_coeffs=BOX_CAST["vel"], dist_pen=BOX_CAST["dist_pen"], coll_coeff=BOX_CAST["coeff"], discrete=False, cast=True)
    else:
        s, rb, static_T, d, traj0 = arm_world()
        kw = dict(vel_coeffs=[1.0], dist_pen=0.025, coll_coeff=20.0, discrete=True, cast=True)      # data/arm_around_table.json:9-21
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb)
    cc.SetContactDistance(d)
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    lg, lo = [], []
    xg, cg, sg = S.optimize(S.TrajProblem(traj0, linearize=S.hip_linearizer(cc), **kw), log=lg)
    xo, co, so = S.optimize(S.TrajProblem(traj0, linearize=S.oracle_linearizer(w, rb, static_T), **kw), log=lo)
    assert sg == so and len(lg) == len(lo), "the two runs took different SQP paths"
    assert cg < lg[0][2] * 0.9, "SQP did not reduce the cost"
    # north-star: final trajectories within the same tolerance as the linearization (1e-5)
    assert np.abs(xg - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max()), np.abs(xg - xo).max()
    assert cg == pytest.approx(co, rel=1e-5, abs=1e-7)
    # the acceptance check of the reference's tests, on the device: the optimised trajectory is collision free where the oracle says so
    hits_g = cc.eng.continuous_check_trajectory(xg, 1, len(xg))
    hits_o = oracle_sweep_hits(w, s, rb, static_T, xg)
    assert len(hits_g) == len(hits_o)
    if name == "box_cast":
        assert len(hits_g) == 0                                                          # cast-cost-unit.cpp:77
    w.close()
    cc.eng.close()
