"""The other queries at full size, checked IN FULL against the CPU oracle (config 4's world: full-body robot + 4096 hulls of
42 vertices; tests/test_gpu_parity.py compares samples, this script everything):

  * RayCastCollision: N rays at R robot configurations -- every closest-hit fraction bit for bit (the oracle casts each ray
    against all 4141 geometries, so the rays are spread over worker processes; none of them touches the GPU);
  * AllVsAll at the same configurations: every contact bit for bit.

  python scripts/check_fullsize_queries.py [--rays 20000] [--configs 3] [--workers 14] [--out gpurun_out/check_queries.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ray_worker(args):
    objT, a, b = args
    from oracle import bindings as B
    from trajopt_amd import synth
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0)
    w = B.oracle_world(s, cap=16)
    out = w.ray_cast(objT, a, b)
    w.close()
    return out


def ava_worker(args):
    objT, d = args
    from oracle import bindings as B
    from trajopt_amd import synth
    s, rb, static_T, _ = synth.config_fullbody_soup(npts=0)
    w = B.oracle_world(s, cap=1 << 20)
    w.set_contact_distance(d)
    out = w.all_vs_all(objT)
    w.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=20000)
    ap.add_argument("--configs", type=int, default=3)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "check_fullsize_queries.json"))
    a_ = ap.parse_args()
    from tests import util
    from trajopt_amd import collision_checker as CC, synth
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0)
    cc = CC.CreateCollisionChecker(s, static_T)
    cc.SetRobot(rb, ignore_zero_state=False)
    cc.SetContactDistance(d)
    eng = cc.eng
    eng.set_capacity(1 << 23, 1 << 21)
    rng = np.random.default_rng(7)
    traj = synth.fullbody_trajectory(rb, rng, 100)
    n = a_.rays
    out = {"rays_per_config": n, "configs": a_.configs, "ray_hits": 0, "ray_mismatches": 0, "all_vs_all_contacts": 0, "all_vs_all_mismatching_configs": 0}
    t0 = time.time()
    with mp.get_context("spawn").Pool(a_.workers) as pool:
        for ci in range(a_.configs):
            q = traj[int(rng.integers(0, 100))]
            fk = eng.forward_kinematics(q[None])[0]
            objT = static_T.copy()
            for i, o in enumerate(rb.link_obj):
                if o >= 0:
                    objT[o] = fk[i]
            a = np.column_stack([rng.uniform(-0.5, 4.0, n), rng.uniform(-1.8, 1.8, n), rng.uniform(1.5, 2.5, n)])
            b = np.column_stack([a[:, 0] + rng.uniform(-0.5, 0.5, n), a[:, 1] + rng.uniform(-0.5, 0.5, n), rng.uniform(-1.0, 0.2, n)])
            a[:200] = objT[rng.integers(0, len(objT), 200), 4:7]          # rays that start at object origins (inside them)
            b[200:220] = a[200:220]                                        # zero-length rays
            fr = eng.ray_cast_poses(objT, a, b)
            cuts = np.linspace(0, n, a_.workers + 1).astype(int)
            parts = pool.map(ray_worker, [(objT, a[cuts[k]:cuts[k + 1]], b[cuts[k]:cuts[k + 1]]) for k in range(a_.workers) if cuts[k + 1] > cuts[k]])
            fo = np.concatenate(parts)
            bad = int((fr.view(np.uint32) != fo.view(np.uint32)).sum())
            out["ray_mismatches"] += bad
            out["ray_hits"] += int((fo < 1).sum())
            ava = eng.all_vs_all_poses(objT)
            oava = pool.apply(ava_worker, ((objT, d),))
            try:
                util.assert_same_contacts(util.records_to_contacts(ava), oava, "all-vs-all config %d" % ci)
            except AssertionError as ex:
                out["all_vs_all_mismatching_configs"] += 1
                print("ALL-VS-ALL MISMATCH:", str(ex)[:300], flush=True)
            out["all_vs_all_contacts"] += len(oava)
            print("[%.0f s] config %d: %d rays (%d hits, %d fractions differ), %d all-vs-all contacts" % (time.time() - t0, ci, n, int((fo < 1).sum()), bad, len(oava)), flush=True)
    eng.close()
    out["seconds"] = time.time() - t0
    out["what"] = "config 4's world (4141 geometries): every ray's closest-hit fraction and every AllVsAll contact vs the CPU oracle, bit for bit"
    with open(a_.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out), flush=True)
    sys.exit(0 if out["ray_mismatches"] == 0 and out["all_vs_all_mismatching_configs"] == 0 else 1)


if __name__ == "__main__":
    main()
