"""ContinuousCheckTrajectory at full size against the CPU oracle (config 4's world: full-body robot, 4096 hulls of 42
vertices): the device answers a whole 100-step trajectory in one call (k_sweep, with its own cull); the oracle has no cull
-- every link geometry is swept against all 4096 hulls -- so S sampled segments are checked, ALL link geometries of each:
hit set, fraction, normal, point and hit geometry bit for bit (oracle in portable-trig mode, as in the tests).
tests/test_gpu_parity.py compares two segments for five links.

  python scripts/check_sweeps_fullsize.py [segments per 2 problems=28] [workers=14] [out.json] [problems=2]      (198 = every segment)
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    segs, static_T, objs, poses, d = args     # poses: {seg: [2][len(objs)][7]}
    from oracle import bindings as B
    from trajopt_amd import synth
    s, rb, _, _ = synth.config_fullbody_soup(npts=0)
    w = B.oracle_world(s, cap=1 << 16)
    w.set_contact_distance(d)
    w.set_trig_mode(1)
    out = {}
    for seg in segs:
        hits = w.continuous_check(static_T, objs, poses[seg])
        out[seg] = sorted((int(h["obj"]), int(h["child"]), int(h["hitObj"]), int(h["hitChild"]), float(h["fraction"]),
                           tuple(np.asarray(h["n"], np.float64).tolist()), tuple(np.asarray(h["pt"], np.float64).tolist())) for h in hits)
    w.set_trig_mode(0)
    w.close()
    return out


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "check_sweeps_fullsize.json")
    NP = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    from trajopt_amd import collision_checker as CC, synth
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0)
    cc = CC.CreateCollisionChecker(s, static_T)
    cc.SetRobot(rb)
    cc.SetContactDistance(d)
    eng = cc.eng
    rng = np.random.default_rng(5)
    tot = {"segments": 0, "hits": 0, "mismatching_segments": []}
    t0 = time.time()
    with mp.get_context("spawn").Pool(W) as pool:
        for p in range(NP):
            traj = synth.fullbody_trajectory(rb, np.random.default_rng(1000 + p), 100)
            sw = eng.continuous_check_trajectory(traj, 1, 100)
            fk = eng.forward_kinematics(traj)
            links = [i for i, o in enumerate(rb.link_obj) if o >= 0]
            objs = [int(rb.link_obj[i]) for i in links]
            segs = sorted(set(int(x) for x in rng.choice(99, min(S // 2, 99), replace=False)) | set(int(t) for t in sw["t"][:2]))
            poses = {seg: fk[seg:seg + 2][:, links].copy() for seg in segs}
            jobs = [(segs[k::W], static_T, objs, {sg: poses[sg] for sg in segs[k::W]}, d) for k in range(W) if segs[k::W]]
            res = {}
            for r in pool.map(worker, jobs):
                res.update(r)
            for seg in segs:
                m = sw[sw["t"] == seg]
                mine = sorted((int(r["idA"]), int(r["childA"]), int(r["idB"]), int(r["childB"]), float(np.float32(r["time"])), tuple(r["n"].tolist()), tuple(r["ptA"].tolist())) for r in m)
                if mine != res[seg]:
                    tot["mismatching_segments"].append({"problem": p, "segment": seg, "engine": len(mine), "oracle": len(res[seg])})
                tot["segments"] += 1
                tot["hits"] += len(res[seg])
            print("[%.0f s] problem %d: %d segments checked, %d hits so far, %d mismatching" % (time.time() - t0, p, tot["segments"], tot["hits"], len(tot["mismatching_segments"])), flush=True)
    eng.close()
    tot["seconds"] = time.time() - t0
    tot["what"] = "ContinuousCheckTrajectory in config 4's world: sampled segments, all link geometries vs all 4096 hulls in the oracle; hits bit for bit"
    json.dump(tot, open(out_path, "w"), indent=1)
    print(json.dumps(tot), flush=True)
    sys.exit(1 if tot["mismatching_segments"] else 0)


if __name__ == "__main__":
    main()
