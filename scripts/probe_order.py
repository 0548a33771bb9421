"""Timing experiment: the two broadphase launches of one linearization in either order (run under rocprofv3 --kernel-trace)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(1 << 23, 1 << 21)
P = 8
dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000 + g), 100) for g in range(P)])
eng.stage_dofs(dofs, P, 100)
order = sys.argv[1] if len(sys.argv) > 1 else "dc"
for it in range(6):
    for i, ch in enumerate(order):
        if ch == "d": eng.enqueue_links_vs_all(-1, True, append=i > 0)
        else: eng.enqueue_cast_vs_all(1, True, append=i > 0)
    eng.sync()
print(eng.counters())
