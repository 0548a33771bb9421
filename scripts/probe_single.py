"""ONE problem's linearization (bench.py's default workload, problem 0), enqueue + fetch, repeated: prints the host-side
phases; run under `rocprofv3 --kernel-trace` + `scripts/rocpd_summary.py timeline` for the launches of one repetition."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s, rb, static_T, d = synth.config_fullbody_soup(npts=npts)
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(1 << 23, 1 << 21)
dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000), 100)])
eng.stage_dofs(dofs, 1, 100)
lat = []
for it in range(30):
    t0 = time.perf_counter()
    eng.enqueue_linearization(-1, 1, True); t1 = time.perf_counter()
    eng.sync(); t2 = time.perf_counter()
    rec, g = eng.fetch(copy=False); t3 = time.perf_counter()
    if it >= 5:
        lat.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, eng.counters()["ms_last_call"]))
m = np.median(np.array(lat), axis=0)
print("enqueue %.3f  sync %.3f  fetch %.3f | total %.3f ms, device %.3f ms, n=%d" % (tuple(m) + (len(rec),)), flush=True)
