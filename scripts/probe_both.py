import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from trajopt_amd import synth, collision_checker as CC, model as M
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
e = cc.eng
e.set_capacity(1 << 22, 1 << 20)
B, T = 6, 40
dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(70 + b), T) for b in range(B)])
e.stage_dofs(dofs, B, T)
for mask, gap in ((-1, 1), (M.GROUP_KINBODY, 2)):
    e.enqueue_links_vs_all(mask, True, append=False)
    e.enqueue_cast_vs_all(gap, True, append=True)
    r0, g0 = e.fetch(); c0 = e.counters()
    e.enqueue_linearization(mask, gap, True)
    r1, g1 = e.fetch(); c1 = e.counters()
    print(len(r0), len(r1), np.array_equal(r0, r1), np.array_equal(g0, g1), [(k, c0[k], c1[k]) for k in ("pair_evals", "gjk_evals", "epa_evals", "contacts", "aabb_tests")])
    assert np.array_equal(r0, r1) and np.array_equal(g0, g1)
    for k in ("pair_evals", "gjk_evals", "epa_evals", "contacts", "aabb_tests"): assert c0[k] == c1[k], k
print("OK")
