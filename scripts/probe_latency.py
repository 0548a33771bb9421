"""Host-side latency breakdown of ONE problem's linearization (config 4), phases timed with perf_counter."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(1 << 23, 1 << 21)
rng = np.random.default_rng(1000)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dofs = np.stack([synth.fullbody_trajectory(rb, rng, 100) for _ in range(P)])
for it in range(12):
    t = [time.perf_counter()]
    eng.stage_dofs(dofs, P, 100); t.append(time.perf_counter())
    eng.enqueue_links_vs_all(-1, True, append=False); t.append(time.perf_counter())
    eng.enqueue_cast_vs_all(1, True, append=True); t.append(time.perf_counter())
    eng.sync(); t.append(time.perf_counter())
    rec, g = eng.fetch(copy=False); t.append(time.perf_counter())
    if it >= 2:
        c = eng.counters()
        print("stage %.3f enq_disc %.3f enq_cast %.3f sync %.3f fetch %.3f | total %.3f ms  device %.3f ms  n=%d pairs=%d" %
              tuple([(t[i + 1] - t[i]) * 1e3 for i in range(5)] + [(t[5] - t[0]) * 1e3, c["ms_last_call"], len(rec), c["pair_evals"]]), flush=True)
