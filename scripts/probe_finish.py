"""k_finish time with and without gradient rows (config 4, P problems): how much of the epilogue is the Jacobian."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(1 << 23, 1 << 21)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(1000)
dofs = np.stack([synth.fullbody_trajectory(rb, rng, 100) for _ in range(P)])
for jac in (True, False, True, False):
    eng.stage_dofs(dofs, P, 100)
    c0 = eng.counters()
    for it in range(5):
        eng.enqueue_links_vs_all(-1, jac, append=False)
        eng.enqueue_cast_vs_all(1, jac, append=True)
        eng.sync()
    c1 = eng.counters()
    print("jac", jac, {k: round((c1[k] - c0[k]) / 10, 4) for k in ("cum_ms_finish", "cum_ms_epa", "cum_ms_gjk", "cum_ms_broadphase")}, flush=True)
