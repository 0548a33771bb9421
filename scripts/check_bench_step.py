"""The bench's own step, checked IN FULL: every record of one timed step of bench.py's default workload (config 4 with
42-vertex hulls, P problems x 100 timesteps, discrete + cast, gradient rows) against the CPU oracle -- contact set and
values bit for bit, rows to 1e-5 relative -- and the pair-evaluation count of the headline metric against the oracle's.
The GPU step takes milliseconds; the oracle needs ~20 core-seconds per problem, so the problems are spread over worker
processes (each rebuilds the scene on the CPU; none of them touches the GPU).

  python scripts/check_bench_step.py [--problems 64] [--workers 14] [--config fullbody42] [--out gpurun_out/check.json]
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


def oracle_worker(args):
    config, probs, dofs, allowed, d, rec_by_problem = args
    from oracle import bindings as B
    from tests import util
    from trajopt_amd import synth
    if config == "multistart":
        s, rb, static_T, _ = synth.config_benchmark_scene("kitchen_counter")
    else:
        s, rb, static_T, _ = synth.config_fullbody_soup(npts=0 if config == "fullbody42" else 120)
    w = B.oracle_world(s, cap=1 << 18)
    w.set_contact_distance(d)
    w.set_allowed(allowed)
    out = {"problems": 0, "records": 0, "pair_evals": 0, "bad": []}
    for b in probs:
        rec, g = rec_by_problem[b]
        disc, cast = rec["cctype"] == 0, rec["cctype"] != 0
        try:
            c0 = w.counter(0)
            orec, og = w.linearize(rb, static_T, dofs[b])
            util.assert_same_linearization(rec[disc], g[disc], orec, og, "problem %d discrete" % b)
            orec, og = w.linearize(rb, static_T, dofs[b], cast=True)
            util.assert_same_linearization(rec[cast], g[cast], orec, og, "problem %d cast" % b)
            out["pair_evals"] += w.counter(0) - c0
        except AssertionError as ex:
            out["bad"].append({"problem": int(b), "what": str(ex)[:300]})
        out["problems"] += 1
        out["records"] += len(rec)
    w.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--config", default="fullbody42", choices=["fullbody42", "fullbody", "multistart"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "check_bench_step.json"))
    a = ap.parse_args()
    import bench
    t0 = time.time()
    s, rb, static_T, d, cc, dofs = bench.build_world(a.config, 0, 1, a.problems)
    eng = cc.eng
    P, T = dofs.shape[0], dofs.shape[1]
    eng.stage_dofs(dofs, P, T)
    eng.enqueue_linearization(-1, 1, True)
    rec, g = eng.fetch()
    c = eng.counters()
    allowed = np.array(cc.allowed, np.uint8)
    print("engine: %d problems x %d timesteps, %d records, %d pair evaluations, %.3f ms on the device (%.1f s with the set-up)"
          % (P, T, len(rec), c["pair_evals"], c["ms_last_call"], time.time() - t0), flush=True)
    order = np.argsort(rec["problem"], kind="stable")
    rec, g = rec[order], g[order]
    cuts = np.searchsorted(rec["problem"], np.arange(P + 1))
    by_problem = {b: (rec[cuts[b]:cuts[b + 1]].copy(), g[cuts[b]:cuts[b + 1]].copy()) for b in range(P)}
    eng.close()
    jobs = [(a.config, list(range(k, P, a.workers)), dofs, allowed, d, {b: by_problem[b] for b in range(k, P, a.workers)}) for k in range(a.workers)]
    jobs = [j for j in jobs if j[1]]
    t1 = time.time()
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        res = []
        for r in pool.imap_unordered(oracle_worker, jobs):
            res.append(r)
            print("[%.0f s] %d of %d workers done, %d problems checked so far" % (time.time() - t1, len(res), len(jobs), sum(x["problems"] for x in res)), flush=True)
    bad = [b for r in res for b in r["bad"]]
    out = {"config": a.config, "problems": P, "timesteps": int(T), "records_compared": int(sum(r["records"] for r in res)),
           "pair_evals_engine": int(c["pair_evals"]), "pair_evals_oracle": int(sum(r["pair_evals"] for r in res)),
           "gjk_evals": int(c["gjk_evals"]), "epa_evals": int(c["epa_evals"]), "mismatching_problems": bad, "oracle_seconds": time.time() - t1,
           "what": "one step of bench.py's workload: every record vs the CPU oracle (contact set + values bit for bit, gradient rows to 1e-5 relative)"}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out), flush=True)
    ok = not bad and out["records_compared"] == len(rec) and (bad or out["pair_evals_engine"] == out["pair_evals_oracle"])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
