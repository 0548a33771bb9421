"""Where the time of tohc_fetch_hinges goes (run on the GPU box; under rocprofv3 --kernel-trace --stats for the kernels)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401  (first, so that its ROCm libraries find the GPU)
from trajopt_amd import collision_checker as CC, synth

P = 64
s, rb, static_T, d = synth.config_fullbody_soup()
dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000 + g), 100) for g in range(P)])
cc = CC.HIPCollisionChecker(s, static_T, None)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(max(1 << 23, P * (1 << 16)), max(1 << 21, P * 40000))
eng.stage_dofs(dofs, P, 100)
for it in range(4):
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    t0 = time.perf_counter(); eng.sync(); t1 = time.perf_counter()
    eng.fetch_hinges_begin(0.05, 0.05); t2 = time.perf_counter()
    h = eng.fetch_hinges_end(copy=False); t3 = time.perf_counter()
    print("sync %.2f ms  begin %.2f ms  end %.2f ms  rows %d nnz %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, h["n_rows"], h["nnz"]), flush=True)
eng.close()
# the pipelined pattern of bench.py, call by call
cc = CC.HIPCollisionChecker(s, static_T, None)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(max(1 << 23, P * (1 << 16)), max(1 << 21, P * 40000))
eng.stage_dofs(dofs, P, 100)
eng.enqueue_links_vs_all(-1, True, append=False)
eng.enqueue_cast_vs_all(1, True, append=True)
for it in range(6):
    t0 = time.perf_counter(); eng.fetch_hinges_begin(0.05, 0.05)
    t1 = time.perf_counter(); eng.stage_dofs(dofs, P, 100)
    t2 = time.perf_counter(); eng.enqueue_links_vs_all(-1, True, append=False); eng.enqueue_cast_vs_all(1, True, append=True)
    t3 = time.perf_counter(); h = eng.fetch_hinges_end(copy=False)
    t4 = time.perf_counter()
    print("pipelined: begin %.2f  stage %.2f  enqueue %.2f  end %.2f  total %.2f ms" % tuple(x * 1e3 for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)), flush=True)
eng.sync()
eng.close()
