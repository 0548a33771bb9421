"""Small driver for rocprofv3: a few config-4 linearizations (discrete + cast) so per-kernel stats can be read."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s, rb, static_T, d = synth.config_fullbody_soup()
eng = engine.Engine(0)
eng.upload_scene(s, static_T); eng.upload_robot(rb); eng.set_contact_distance(d)
eng.set_capacity(1 << 23, 1 << 21)
rng = np.random.default_rng(2)
dofs = np.stack([synth.fullbody_trajectory(rb, rng) for _ in range(B)])
for r in range(reps):
    eng.links_vs_all_batch(dofs, B, 100, copy=False)
    c = eng.counters()
    eng.cast_vs_all_batch(dofs, B, 100, 1, copy=False)
    print(r, c, eng.counters(), flush=True)
