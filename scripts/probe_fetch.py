import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth, engine as E
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(1 << 23, 1 << 21)
rng = np.random.default_rng(1000)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dofs = np.stack([synth.fullbody_trajectory(rb, rng, 100) for _ in range(P)])
for it in range(6):
    eng.stage_dofs(dofs, P, 100)
    eng.enqueue_links_vs_all(-1, True, append=False)
    eng.enqueue_cast_vs_all(1, True, append=True)
    eng.sync()
    rec, g, n = C.c_void_p(), E._f64p(), C.c_int()
    t0 = time.perf_counter()
    eng.lib.tohc_fetch(eng.h, C.byref(rec), C.byref(g), C.byref(n))
    t1 = time.perf_counter()
    r = eng._collect(rec, g, n, eng.K, False)
    t2 = time.perf_counter()
    print("C fetch %.3f ms, python collect %.3f ms, n=%d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, n.value), flush=True)
