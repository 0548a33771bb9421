import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from trajopt_amd import engine, model as M, synth
s, rb, static_T, d = synth.config_fullbody_soup()
eng = engine.Engine(0)
eng.upload_scene(s, static_T); eng.upload_robot(rb); eng.set_contact_distance(d)
eng.set_capacity(1<<23, 1<<20)
rng = np.random.default_rng(2)
for B in (1, 1, 4, 16):
    dofs = np.stack([synth.fullbody_trajectory(rb, rng) for _ in range(B)])
    for rep in range(3):
        t0 = time.perf_counter()
        rec, g = eng.links_vs_all_batch(dofs, B, 100, filter_mask=-1, copy=False)
        t1 = time.perf_counter()
        c = eng.counters()
        recc, gc = eng.cast_vs_all_batch(dofs, B, 100, 1, copy=False)
        t2 = time.perf_counter()
        cc = eng.counters()
    print("B", B, "disc: wall %.3f ms dev %.3f ms" % ((t1-t0)*1e3, c['ms_last_call']), c, "| cast: wall %.3f ms dev %.3f" % ((t2-t1)*1e3, cc['ms_last_call']), cc, flush=True)
    print("   cctypes", np.bincount(recc['cctype'], minlength=4), "dist min", rec['dist'].min() if len(rec) else None)
