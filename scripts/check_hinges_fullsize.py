"""tohc_fetch_hinges at the bench's size against the host assembly (collision_checker.bpmpd_collision_block: what sco's
CollisionsToDistanceExpressions -> convex -> addHinge -> BPMPDModel::optimize build from the fetched slab): every array of
the block BIT-identical.  tests/test_gpu_parity.py::test_hinge_block_equals_host_assembly does this for 3 problems x 9
timesteps; here P problems of bench.py's default workload (config 4, 100 timesteps).

  python scripts/check_hinges_fullsize.py [problems=64] [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                               # noqa: E402
from trajopt_amd import collision_checker as CC            # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "check_hinges_fullsize.json")
s, rb, static_T, d, cc, dofs = bench.build_world("fullbody42", 0, 1, P)
eng = cc.eng
T = dofs.shape[1]
eng.stage_dofs(dofs, P, T)
eng.enqueue_linearization(-1, 1, True)
rec, g = eng.fetch()
h = eng.fetch_hinges(0.05, 0.03)
t0 = time.time()
ref = CC.bpmpd_collision_block(rec, g, P, T, rb.K, 1, 0.05, 0.03)
dt = time.time() - t0
bad = [k for k in ("B", "T", "K", "gap", "n_rows", "nnz") if h[k] != ref[k]]
bad += [k for k in ("row_off", "rhs", "dist", "row_t", "col_off", "col_idx", "col_val") if h[k].dtype != ref[k].dtype or h[k].tobytes() != ref[k].tobytes()]
out = {"problems": P, "timesteps": int(T), "rows": int(h["n_rows"]), "nonzeros": int(h["nnz"]), "records": int(len(rec)), "host_assembly_seconds": dt,
       "arrays_that_differ": bad, "what": "tohc_fetch_hinges vs the host assembly of the same slab, every array bit for bit"}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out), flush=True)
eng.close()
sys.exit(1 if bad else 0)
