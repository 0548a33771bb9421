#!/usr/bin/env python3
"""bench.py -- collision-cost linearization throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE.json configs[3] -- full-body PR2-like robot (K=18), 100 timesteps, against a
4096-hull convex soup, d = 0.09 -- batched as `problems_per_gpu` independent multi-start trajectories per GPU
(configs[4]'s sharding: problems are partitioned over ranks, no data-path collective).  One STEP = one SQP
linearization of every problem on the rank: the discrete collision cost (100 terms, self collision included) and the
cast collision cost (99 swept terms), contacts + gradient rows, into one device slab -- FK, broadphase, GJK, EPA and
the Jacobian epilogue all inside the timed region; the DOF vectors are already resident in HBM (tohc_stage_dofs).
`value` = pair-evals of all ranks / max-over-ranks wall time.  The PCIe-inclusive API rate (stage + enqueue + fetch to
pinned host memory, canonical order) is reported beside it as `api_value`; it is never `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per launch from the PMC counters: profiles/r01_h_hbm_traffic_pmc.csv, collected on this workload (64
# problems per GPU, scripts/profile_bench.sh) as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in separate
# passes, bytes = (2*FETCH + WRITE) * 1024), mean of the discrete and the cast launch; "epa" = k_pen + k_epa.  Far below
# the algorithmic bytes for the broadphase and GJK kernels because the 2 MB of hull vertices and the box table are
# re-read from L2 / Infinity Cache, not HBM.
TRAFFIC_BYTES_PER_LAUNCH = {"broadphase": 282.3e6, "gjk": 44.6e6, "epa": 42.7e6, "finish": 385.6e6}


def build_world(rank, world, problems, T=100):
    """`problems` multi-start trajectories per GPU (weak scaling): the job's problems*world problems are numbered
    job-wide, problem g is seeded with 1000+g, and rank r takes the block multi_gpu.partition gives it."""
    from trajopt_amd import collision_checker as CC, multi_gpu, synth
    s, rb, static_T, d = synth.config_fullbody_soup()
    cc = CC.HIPCollisionChecker(s, static_T, None, device=rank_device(rank))
    cc.SetRobot(rb)                       # uploads + IgnoreZeroStateSelfCollisions, like the reference's AddKinBody path
    cc.SetContactDistance(d)
    cc.eng.set_capacity(max(1 << 23, problems * (1 << 16)), max(1 << 21, problems * 40000))   # ~30 K GJK pairs / 25 K records per problem and step
    lo, hi = multi_gpu.partition(problems * world, world, rank)
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000 + g), T) for g in range(lo, hi)])
    return s, rb, static_T, d, cc, dofs


def rank_device(rank):
    return int(os.environ.get("LOCAL_RANK", rank))


def cpu_baseline(s, rb, static_T, d, allowed, trajs, budget_s=12.0, only=None):
    """The CPU path on ONE host core, on a bounded sample of the same workload: timesteps of the rank's problems, each
    problem taken with a stride of 7, problem after problem until ~budget_s of work.  kind=reference: the reference's own vendored Bullet (oracle/_ref) doing
    LinksVsAll + CastVsAll on poses from the oracle's FK (FK / Jacobian time NOT charged to it);
    kind=port: the C restatement doing the full linearization."""
    from oracle import bindings as B
    T = trajs.shape[1]
    order = [(p, t) for p in range(len(trajs)) for t in list(range(0, T - 1, 7)) + [t for t in range(T - 1) if t % 7]]
    out = {}
    wo = B.oracle_world(s, cap=1 << 18)
    wo.set_contact_distance(d)
    wo.set_allowed(allowed)
    # port
    t0 = time.perf_counter()
    pe0 = wo.counter(0)
    used = 0
    for p, t in (order if only != "reference" else []):
        wo.linearize(rb, static_T, trajs[p, t:t + 1])
        wo.linearize(rb, static_T, trajs[p, t:t + 2], cast=True)
        used += 1
        if time.perf_counter() - t0 > budget_s / 2:   # the port is the secondary figure
            break
    dt = time.perf_counter() - t0
    out["port"] = dict(value=(wo.counter(0) - pe0) / dt, seconds=dt, timesteps=used)
    if only == "port":
        wo.close()
        return out
    if B.ref_available():
        wr = B.ref_world(s, cap=1 << 18)
        wr.set_contact_distance(d)
        wr.set_allowed(allowed)
        lib = wr.lib
        objs, links = rb.affected_links()
        link_obj = np.array(rb.link_obj)
        have = link_obj >= 0

        def poses(q):
            Tall = static_T.copy()
            fk = B.oracle_fk(rb, q)
            Tall[link_obj[have]] = fk[have]
            return Tall, fk

        g0 = lib.ref_gjk_checks()
        tsum, used = 0.0, 0
        for p, t in order:
            P0, fk0 = poses(trajs[p, t])
            _, fk1 = poses(trajs[p, t + 1])
            t0 = time.perf_counter()
            wr.links_vs_all(P0, objs)
            wr.cast_vs_all(P0, objs, fk0[links], fk1[links])
            tsum += time.perf_counter() - t0
            used += 1
            if tsum > budget_s:
                break
        out["reference"] = dict(value=(lib.ref_gjk_checks() - g0) / tsum, seconds=tsum, timesteps=used)
        wr.close()
    wo.close()
    return out


def cpu_worker(idx, n, problems, budget_s):
    """One of the n worker processes of the all-cores CPU baseline: the same bounded sample, on the problems idx, idx+n, ..."""
    from trajopt_amd import multi_gpu, synth
    from oracle import bindings as B
    s, rb, static_T, d = synth.config_fullbody_soup()
    allowed = np.load(os.environ["BENCH_ALLOWED"])
    mine = list(range(idx, problems, n)) or [idx % problems]
    trajs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(1000 + g), 100) for g in mine])
    cb = cpu_baseline(s, rb, static_T, d, allowed, trajs, budget_s=budget_s, only="reference" if B.ref_available() else "port")
    k = "reference" if "reference" in cb else "port"
    print(json.dumps({"kind": k, "pair_evals": cb[k]["value"] * cb[k]["seconds"], "seconds": cb[k]["seconds"]}), flush=True)


def cpu_baseline_all_cores(allowed, problems, budget_s=10.0):
    """The same CPU path on every host core the job may use (the box gives a one-GPU job 16): independent worker
    processes (children started with subprocess, each its own interpreter; none touches the GPU), one per core, each
    over its own problems for ~budget_s; rate = summed pair-evals / longest worker time."""
    import subprocess
    import tempfile
    n = max(1, min(16, len(os.sched_getaffinity(0))))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "allowed.npy")
        np.save(path, allowed)
        env = dict(os.environ, BENCH_ALLOWED=path, OMP_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), str(n), str(problems), str(budget_s)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(n)]
        outs = [p.communicate(timeout=180)[0] for p in procs]
    res = [json.loads(o.strip().splitlines()[-1]) for o, p in zip(outs, procs) if p.returncode == 0 and o.strip()]
    if not res:
        return None
    return {"value": sum(r["pair_evals"] for r in res) / max(r["seconds"] for r in res), "cores": len(res), "kind": res[0]["kind"]}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--problems-per-gpu", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hot-path-only", action="store_true",
                    help="skip the side measurements (API rate, single-problem latency, ContinuousCheckTrajectory) so that every launch "
                         "a profiler sees belongs to the timed workload: the command the profiles/ summaries are made with")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(rank_device(rank))
        dist.init_process_group("nccl", device_id=torch.device("cuda", rank_device(rank)))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the engine has no CPU fallback"

    P, T = args.problems_per_gpu, 100
    s, rb, static_T, d, cc, dofs = build_world(rank, world, P, T)
    eng = cc.eng

    cast_first = os.environ.get("BENCH_CAST_FIRST", "0") == "1"   # timing experiment: which pass is enqueued first

    def step():
        if cast_first:
            eng.enqueue_cast_vs_all(1, True, append=False)
            eng.enqueue_links_vs_all(-1, True, append=True)
        else:
            eng.enqueue_links_vs_all(-1, True, append=False)
            eng.enqueue_cast_vs_all(1, True, append=True)
        eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.stage_dofs(dofs, P, T)
    for _ in range(args.warmup):
        step()
    c0 = eng.counters()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    rec, grads = eng.fetch()

    pair_evals = c1["cum_pair_evals"] - c0["cum_pair_evals"]
    from trajopt_amd import multi_gpu
    dt_max, pairs_all = multi_gpu.reduce_measurement(dt, pair_evals, dist, device="cuda")

    # The north-star's one exchange step, AFTER the path and outside `value`: every rank's result slab of one step to
    # every rank (RCCL all-gather over xGMI; padded fixed-size slabs), so a single SQP host can read all of them.
    gather_ms = gather_bytes = None
    if dist is not None:
        multi_gpu.all_gather_slabs(rec[:1024], grads[:1024], dist, device="cuda")   # warm-up (communicator set-up)
        barrier()
        tg = time.perf_counter()
        recs_all, _ = multi_gpu.all_gather_slabs(rec, grads, dist, device="cuda")
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        gather_bytes = int(sum(len(r) for r in recs_all) * (152 + 16 * rb.K))
        assert len(recs_all) == world and len(recs_all[rank]) == len(rec)

    # PCIe-inclusive API rate on rank 0's data (never `value`)
    api = lat_ms = lat_dev_ms = sweep_ms = n_sw = None
    if rank == 0 and not args.hot_path_only:
        ta = time.perf_counter()
        n_api = 3
        pe = 0
        for _ in range(n_api):
            eng.stage_dofs(dofs, P, T)
            eng.enqueue_links_vs_all(-1, True, append=False)
            eng.enqueue_cast_vs_all(1, True, append=True)
            eng.fetch(copy=False)
            pe += eng.counters()["pair_evals"]
        api = pe / (time.perf_counter() - ta)
        # latency of ONE problem's linearization, results in host memory (north-star: < 1 ms)
        eng.stage_dofs(dofs[:1], 1, T)
        lat = []
        for _ in range(10):
            tl = time.perf_counter()
            eng.enqueue_links_vs_all(-1, True, append=False)
            eng.enqueue_cast_vs_all(1, True, append=True)
            eng.fetch(copy=False)
            lat.append((time.perf_counter() - tl) * 1e3)
        lat_ms = float(np.median(lat))
        lat_dev_ms = eng.counters()["ms_last_call"]
        # ContinuousCheckTrajectory (the acceptance check of the reference's tests) of one problem, hits in host memory
        sw = []
        for _ in range(5):
            tl = time.perf_counter()
            n_sw = len(eng.continuous_check_trajectory(dofs[0], 1, T))
            sw.append((time.perf_counter() - tl) * 1e3)
        sweep_ms = float(np.median(sw))

    if rank == 0:
        launches = c1["cum_launches"] - c0["cum_launches"]
        ms = {k: (c1["cum_ms_" + k] - c0["cum_ms_" + k]) / max(launches, 1) for k in ("fk", "children", "broadphase", "gjk", "epa", "finish")}
        # roofline: ALGORITHMIC bytes per launch of each kernel / its HIP-event duration;
        # the entry for the kernel with the longest duration is reported as `roofline`.
        #   shape side = 12 B per hull vertex (a box: its 3 half extents), pose = 48 B (fp32 3x4), cull box = 24 B
        a = s.arrays
        child_v = np.where(a["kind"][a["child_shape"]] == 1, a["vcnt"][a["child_shape"]], 1)
        objs, _ = rb.affected_links()
        qch = np.concatenate([np.arange(a["obj_off"][o], a["obj_off"][o] + a["obj_cnt"][o]) for o in objs])
        n_q = float(child_v[qch].mean())
        n_x = float(child_v.mean())       # dominated by the 4096 soup children
        L = max(launches, 1)
        pairs_l = pair_evals / L
        gjk_l = (c1["cum_gjk_evals"] - c0["cum_gjk_evals"]) / L
        epa_l = (c1["cum_epa_evals"] - c0["cum_epa_evals"]) / L
        hits_l = c1["contacts"] / 2.0     # the slab of one step holds both launches
        units_l = P * (T - 0.5) * len(qch)  # (unit, query child) wavefronts per launch, mean of the discrete and the cast pass
        side = lambda nv: 12.0 * nv + 48.0
        alg = {
            # every pair evaluation reads the other child's box, shape and pose; each wavefront its query child's
            "broadphase": pairs_l * (24.0 + side(n_x)) + units_l * (24.0 + side(n_q)) + gjk_l * 16.0,
            "gjk": gjk_l * (16.0 + side(n_q) + side(n_x)) + hits_l * 32.0 + epa_l * 36.0,
            "epa": epa_l * (36.0 + side(n_q) + side(n_x) + 32.0),
            "finish": hits_l * (32.0 + 16.0 + 2 * 7 * 8.0 * 8 + 152.0 + 16.0 * rb.K),   # hit + pair + two link chains (~8 fp64 poses) + record + gradient rows
        }
        roof_all = {k: {"algorithmic_bytes_per_launch": alg[k], "ms_per_launch": ms[k],
                        "achieved_GBs": (alg[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0)} for k in alg}
        dom = max(alg, key=lambda k: ms[k])
        achieved = roof_all[dom]["achieved_GBs"]
        kname = {"broadphase": "k_broadphase", "gjk": "k_gjk", "epa": "k_pen+k_epa", "finish": "k_finish"}[dom]
        out = {
            "metric": "signed-distance+Jacobian pair-evals/s (node)",
            "value": pairs_all / dt_max,
            "unit": "pair-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[3]: full-body PR2-like (K=18) x 100 timesteps vs 4096-hull convex soup, d=0.09, "
                                   "discrete+cast collision costs with gradient rows; %d independent multi-start problems per GPU" % P,
                       "problems_per_gpu": P, "timesteps": T, "K": rb.K, "parallelism": "problems sharded over %d GPU(s), no collective" % world},
            "pair_evals_per_step_per_gpu": pair_evals / args.steps,
            "gjk_evals_per_step_per_gpu": (c1["cum_gjk_evals"] - c0["cum_gjk_evals"]) / args.steps,
            "epa_evals_per_step_per_gpu": (c1["cum_epa_evals"] - c0["cum_epa_evals"]) / args.steps,
            "contacts_per_step_per_gpu": int(c1["contacts"]),
            "ms_per_sqp_linearization_single_problem_host_delivered": lat_ms,
            "ms_per_sqp_linearization_single_problem_device": lat_dev_ms,
            "api_value": api,
            "allgather_slabs_ms": gather_ms, "allgather_slabs_bytes": gather_bytes,
            "ms_continuous_check_trajectory_single_problem": sweep_ms,
            "continuous_check_hits_single_problem": n_sw,
            "kernel_ms_per_launch": ms,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": TRAFFIC_BYTES_PER_LAUNCH.get(dom),
                         "note": "dominant kernel by HIP-event duration; dependent-latency bound (LDS / L2 round trips of an "
                                 "iterative fp32 algorithm), far below the HBM roof by nature; bytes are compulsory only (SURVEY 8d)"},
            "roofline_all": roof_all,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 at N = 1 only
            cb = cpu_baseline(s, rb, static_T, d, cc.allowed, dofs)
            kind = "reference" if "reference" in cb else "port"
            out["cpu_baseline"] = {"value": cb[kind]["value"], "unit": "pair-evals/s", "cores": 1, "kind": kind,
                                   "sample": "%d (problem, timestep) units of this rank's workload (%.1f s of CPU work): discrete LinksVsAll + CastVsAll per unit on one core"
                                             % (cb[kind]["timesteps"], cb[kind]["seconds"]),
                                   "port_value": cb["port"]["value"]}
            try:   # the same on every host core of the job (north-star: "timed on the box's host cores, with the core count stated")
                out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(cc.allowed, P)
            except Exception as ex:   # a reported extra: never fails the bench
                out["cpu_baseline"]["all_cores"] = {"error": str(ex)[:200]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    cc.eng.close()


if __name__ == "__main__":
    main()
