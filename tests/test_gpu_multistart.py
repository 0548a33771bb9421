"""BASELINE.json configs[4], ONE GPU's share at size: 1024 independent python_examples/benchmark.py-shaped problems
(benchmark.py:71-118: right arm K = 7, T = 11, a discrete and a continuous collision cost, dist_pen 0.05 => d = 0.09) in
the reference's kitchen_counter world (52 boxes), seeds 1000 + g as bench.py --config multistart draws them, through
tohc_enqueue_linearization with the engine's DEFAULT settings -- so the broadphase is k_broadphase_unit (one wavefront per
(problem, timestep); taken from 1024 units on), the route a multi-start job actually runs.

Too many problems for the scalar oracle in the test budget, so: size-independent properties of the whole slab, and 24
sampled problems (first, last, and seeded picks) against the oracle, contacts bit for bit and gradient rows to 1e-5.
"""
import numpy as np
import pytest

from oracle import bindings as B
from tests import util
from trajopt_amd import model as M, synth

pytestmark = pytest.mark.gpu
P, SAMPLES = 1024, 24


def test_multistart_share_at_size_vs_oracle(built):
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_benchmark_scene("kitchen_counter")
    dofs = np.concatenate([synth.benchmark_trajectories(rb, np.random.default_rng(1000 + g), 1) for g in range(P)])
    _, T, K = dofs.shape
    assert (T, K) == (11, 7)
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb)                       # + IgnoreZeroStateSelfCollisions, like the reference's AddKinBody path
    cc.SetContactDistance(d)
    eng = cc.eng
    eng.set_capacity(max(1 << 22, P * (1 << 12)), max(1 << 20, P * 2048))   # bench.py's capacities for this workload
    eng.stage_dofs(dofs, P, T)
    eng.enqueue_linearization(-1, 1, True)
    rec, g = eng.fetch()
    c = eng.counters()
    assert c["contacts"] == len(rec) > P and c["pair_evals"] >= c["gjk_evals"] >= c["epa_evals"] > 0
    assert c["aabb_tests"] > c["pair_evals"]
    # --- size-independent properties of the whole slab ---
    disc = rec["cctype"] == 0
    assert disc.any() and (~disc).any()
    assert (np.diff(rec["problem"]) >= 0).all()                                         # canonical order
    key = np.stack([rec["problem"], rec["t"], (~disc).astype(np.int32), rec["idA"], rec["childA"], rec["idB"], rec["childB"]], 1)
    # one contact per child pair and pass -- except that a robot-robot pair is reported from each of its two query links
    # (m_filterMask = -1 and the loop is over query links): exact duplicates, never more than two, discrete only
    uk, first, cnt = np.unique(key, axis=0, return_index=True, return_counts=True)
    grp = np.asarray(s.arrays["obj_group"])
    assert cnt.max() <= 2
    dup = rec[first[cnt == 2]]
    assert (dup["cctype"] == 0).all() and (grp[dup["idA"]] & grp[dup["idB"]] & M.GROUP_ROBOT).all()
    nxt = rec[first[cnt == 2] + 1]                      # (canonical order puts the twin right behind)
    assert all(np.array_equal(dup[f], nxt[f]) for f in ("dist", "n", "ptA", "ptB", "const0"))
    assert (rec["dist"] <= np.float32(d)).all() and np.allclose(np.linalg.norm(rec["n"], axis=1), 1, atol=1e-5)
    assert (rec["t"][disc] < T).all() and (rec["t"][~disc] < T - 1).all()
    assert (grp[rec["idA"][~disc]] & M.GROUP_KINBODY).all() and (grp[rec["idB"][~disc]] & M.GROUP_ROBOT).all()   # cast: B is the swept link (bcc:913-917)
    q0 = dofs[rec["problem"], rec["t"]]
    q1 = dofs[rec["problem"], np.minimum(rec["t"] + 1, T - 1)]
    lin0 = rec["const0"] + np.einsum("nk,nk->n", g[:, 0], q0)
    lin1 = rec["const1"] + np.einsum("nk,nk->n", g[:, 1], q1)
    assert np.allclose(lin0[disc], rec["dist"][disc], atol=1e-9)                         # const0 + grad0 . q = dist
    tm = rec["time"][~disc].astype(np.float64)
    assert np.allclose((1 - tm) * lin0[~disc] + tm * lin1[~disc], rec["dist"][~disc], atol=1e-9)   # collision_terms.cpp:64-80
    assert not g[disc][:, 1].any()                                                       # a discrete row has no second half
    # determinism: the same enqueue again gives the identical slab; and the two-call route (two broadphase launches) too
    eng.enqueue_linearization(-1, 1, True)
    rec2, g2 = eng.fetch()
    assert rec.tobytes() == rec2.tobytes() and np.array_equal(g, g2)
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    rec3, g3 = eng.fetch()
    assert rec.tobytes() == rec3.tobytes() and np.array_equal(g, g3)
    # problems are independent: a problem alone (the per-query broadphase: 11 units) = its share of the batch
    for p in (0, 517):
        eng.stage_dofs(dofs[p:p + 1], 1, T)
        eng.enqueue_linearization(-1, 1, True)
        r1, g1 = eng.fetch()
        sel = rec["problem"] == p
        ref = rec[sel].copy(); ref["problem"] = 0
        assert r1.tobytes() == ref.tobytes() and np.array_equal(g1, g[sel])
    # --- sampled problems against the oracle ---
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    w.set_allowed(cc.allowed)
    picks = sorted(set([0, P - 1] + np.random.default_rng(5).choice(P, SAMPLES, replace=False).tolist()))[:SAMPLES]
    total = 0
    for p in picks:
        sel = rec["problem"] == p
        for cast in (False, True):
            m = sel & (disc != cast)
            orec, og = w.linearize(rb, static_T, dofs[p], cast=cast)
            util.assert_same_linearization(rec[m], g[m], orec, og, "multistart problem %d cast=%s" % (p, cast))
            total += len(orec)
    assert total > SAMPLES
    w.close()
    eng.close()
