"""Runs config 4 with 42-vertex hulls (bench.py's default workload) on an experiment build of the library (TOHC_LIB or
scripts/native/libtohc_prof.so) for a few steps; a TOHC_EPA_PROF build prints its in-kernel EPA cycle profile at every sync."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajopt_amd import engine
if not os.environ.get("TOHC_LIB"):
    engine.LIB_PATH = os.path.join(ROOT, "scripts", "native", "libtohc_prof.so")
from trajopt_amd import collision_checker as CC, synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s, rb, static_T, d = synth.config_fullbody_soup(npts=0 if len(sys.argv) < 3 else int(sys.argv[2]))
cc = CC.HIPCollisionChecker(s, static_T, None, device=0)
cc.SetRobot(rb); cc.SetContactDistance(d)
eng = cc.eng
eng.set_capacity(max(1 << 23, P * (1 << 16)), max(1 << 21, P * 40000))
dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000 + g), 100) for g in range(P)])
eng.stage_dofs(dofs, P, 100)
for it in range(3):
    eng.enqueue_linearization(-1, 1, True)
    eng.sync()
    c = eng.counters()
    print("step %d: %.3f ms device, pairs %d gjk %d epa %d redone %d contacts %d" % (it, c["ms_last_call"], c["pair_evals"], c["gjk_evals"], c["epa_evals"], c["epa_redone"], c["contacts"]), flush=True)
