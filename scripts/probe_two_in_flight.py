"""Two batches in flight on one GPU (GPU box): bench.py's step on the default workload with ONE engine (enqueue, sync, enqueue,
sync ...: what `value` is) against TWO engines that take turns -- engine B's step is enqueued before engine A's is waited
for, so the GPU always has the next batch's kernels queued behind the tail of the current one.  Each engine holds its own
half of the problems' worth of buffers; the pair evaluations per step are the same in both set-ups.

  python scripts/probe_two_in_flight.py [--problems 64] [--seconds 2] [--out gpurun_out/two_in_flight.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--config", default="fullbody42")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "two_in_flight.json"))
    a = ap.parse_args()
    import torch
    import bench
    assert torch.cuda.is_available()
    worlds = [bench.build_world(a.config, 0, 1, a.problems) for _ in range(2)]
    engs = [w[4].eng for w in worlds]
    for w, e in zip(worlds, engs):
        dofs = w[5]
        e.stage_dofs(dofs, dofs.shape[0], dofs.shape[1])
        for _ in range(3):
            e.enqueue_linearization(-1, 1, True); e.sync()

    def pairs(e):
        return e.counters()["cum_pair_evals"]

    out = {"config": a.config, "problems_per_batch": a.problems}
    # one engine, one batch at a time
    e = engs[0]
    torch.cuda.synchronize()
    p0 = pairs(e); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < a.seconds:
        e.enqueue_linearization(-1, 1, True); e.sync(); n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["one_at_a_time"] = {"steps": n, "ms_per_step": dt / n * 1e3, "pair_evals_per_s": (pairs(e) - p0) / dt}
    # two engines taking turns
    torch.cuda.synchronize()
    p0 = pairs(engs[0]) + pairs(engs[1]); t0 = time.perf_counter(); n = 0
    engs[0].enqueue_linearization(-1, 1, True)
    cur = 0
    while time.perf_counter() - t0 < a.seconds:
        engs[1 - cur].enqueue_linearization(-1, 1, True)
        engs[cur].sync(); n += 1
        cur = 1 - cur
    engs[cur].sync(); n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["two_in_flight"] = {"steps": n, "ms_per_step": dt / n * 1e3, "pair_evals_per_s": (pairs(engs[0]) + pairs(engs[1]) - p0) / dt}
    out["gain"] = out["two_in_flight"]["pair_evals_per_s"] / out["one_at_a_time"]["pair_evals_per_s"]
    rec0, _ = engs[0].fetch()
    rec1, _ = engs[1].fetch()
    out["records_equal"] = bool(rec0.tobytes() == rec1.tobytes())
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out), flush=True)
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
