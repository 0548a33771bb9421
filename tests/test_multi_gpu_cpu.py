"""N > 1 path on the CPU: two `gloo` ranks shard a set of multi-start problems (trajopt_amd/multi_gpu.py), each rank
linearizes its own block -- with the CPU oracle standing in for the GPU engine, this is a test -- and the union must
equal the single-process result; the scalar reductions of the bench contract (max time, summed units) are checked too.
Computing needs no collective; the optional exchange step after it (all_gather_slabs: every rank's slab on every rank,
the north-star's "all-gather of linearizations") is checked bit for bit as well."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajopt_amd import multi_gpu  # noqa: E402

N_PROBLEMS, T = 5, 6   # odd on purpose: the blocks are ragged (3 + 2)


def _problems():
    from trajopt_amd import synth
    s, rb, static_T, d = synth.config_arm_table(seed=0)
    q0 = rb.q_robot[1:8]
    trajs = []
    for g in range(N_PROBLEMS):
        rng = np.random.default_rng(100 + g)
        trajs.append(synth.straight_line(q0, q0 + rng.uniform(-0.8, 0.8, size=7), T))
    return s, rb, static_T, d, np.stack(trajs)


def _linearize(s, rb, static_T, d, trajs, lo):
    from oracle import bindings as B
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    out = []
    for i, tr in enumerate(trajs):
        for cast in (False, True):
            rec, grads = w.linearize(rb, static_T, tr, cast=cast)
            for r, g in zip(rec, grads):
                out.append((lo + i, int(cast), int(r["t"]), int(r["c"]["objA"]), int(r["c"]["childA"]), int(r["c"]["objB"]),
                            int(r["c"]["childB"]), float(r["c"]["dist"]), g.tobytes()))
    pe = w.counter(0)
    w.close()
    return out, pe


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, rb, static_T, d, trajs = _problems()
        lo, hi = multi_gpu.partition(N_PROBLEMS, world, rank)
        mine, pe = _linearize(s, rb, static_T, d, trajs[lo:hi], lo)
        dist.barrier()
        tmax, usum = multi_gpu.reduce_measurement(1.0 + rank, pe, dist)
        counts = multi_gpu.gather_counts(len(mine), dist)
        # the optional exchange step: every rank's slab on every rank (ragged counts, padded messages)
        from trajopt_amd import engine
        rec = np.zeros(len(mine), engine.RECORD_DTYPE)
        rec["problem"] = [m[0] for m in mine]; rec["cctype"] = [m[1] for m in mine]; rec["t"] = [m[2] for m in mine]
        rec["dist"] = [m[7] for m in mine]
        grads = np.stack([np.frombuffer(m[8], np.float64) for m in mine]).reshape(len(mine), 2, -1)
        recs, gs = multi_gpu.all_gather_slabs(rec, grads, dist)
        gathered = [(r["problem"].tolist(), r["dist"].tolist(), g.tobytes()) for r, g in zip(recs, gs)]
        q.put((rank, lo, hi, mine, pe, tmax, usum, counts.tolist(), gathered))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_covers_every_problem_once():
    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            blocks = [multi_gpu.partition(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        multi_gpu.partition(4, 2, 2)


def test_single_process_reduction_is_identity():
    assert multi_gpu.reduce_measurement(0.25, 1000) == (0.25, 1000.0)
    assert multi_gpu.gather_counts(7).tolist() == [7]


@pytest.mark.timeout(300)
def test_two_gloo_ranks_shard_the_problems():
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s, rb, static_T, d, trajs = _problems()
    ref, pe_ref = _linearize(s, rb, static_T, d, trajs, 0)
    assert [(g[1], g[2]) for g in got] == [(0, 3), (3, 5)]
    union = got[0][3] + got[1][3]
    assert sorted(union) == sorted(ref) and len(ref) > 0          # same contacts, distances and gradient rows, bit for bit
    assert got[0][4] + got[1][4] == pe_ref                          # pair evaluations add up: nothing done twice or skipped
    for g in got:
        assert g[5] == 2.0 and g[6] == float(pe_ref)                # max over ranks of the time, sum of the units
        assert g[7] == [len(got[0][3]), len(got[1][3])]
        # all_gather_slabs: both ranks hold both slabs, bit for bit, in rank order
        for r in range(2):
            src = got[r][3]
            assert g[8][r][0] == [m[0] for m in src] and g[8][r][1] == [m[7] for m in src]
            assert g[8][r][2] == b"".join(m[8] for m in src)


@pytest.mark.timeout(300)
def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two rank processes itself (torch.distributed.run, before any
    GPU call); rank 0 prints the one JSON line with n_gpus = 2 and the job-wide unit count.  --dry-run: gloo and a stub
    engine, so this runs without a GPU.  A WORLD_SIZE that contradicts --gpus is refused."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--problems-per-gpu", "5"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["scaling"] == "weak"
    assert out["units_all_ranks"] == 2 * 5 * 1000.0 * 3           # both ranks' blocks of 5 problems took part
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


# ----------------------------------------------------------------------------------------------------------------
# timestep-sharded mode: ONE trajectory, its timesteps block-partitioned over the ranks with a halo of `gap` poses
# (the reference hatches one cost per timestep / per (t, t + gap): problem_description.cpp:516-529)
# ----------------------------------------------------------------------------------------------------------------
T_LONG = 11


def _long_problem():
    from trajopt_amd import synth
    s, rb, static_T, d = synth.config_arm_table(seed=0)
    q0 = rb.q_robot[1:8]
    rng = np.random.default_rng(5)   # (a seed whose sweep meets the table from t = 2 on: cast contacts in both shards)
    return s, rb, static_T, d, synth.straight_line(q0, q0 + rng.uniform(-0.9, 0.9, size=7), T_LONG)


def _window_records(w, rb, static_T, traj, sh, gap):
    """What a shard's engine call returns (tohc_enqueue_linearization_window), with the oracle standing in for the GPU:
    discrete over the first n_discrete staged poses, cast over every staged pair; t counts from the shard's first pose."""
    from trajopt_amd import engine
    staged = traj[sh["lo"]:sh["stage_hi"]]
    parts = []
    for cast in (False, True):
        use = staged if cast else staged[:sh["n_discrete"]]
        if len(use) <= (gap if cast else 0):
            continue
        rec, grads = w.linearize(rb, static_T, use, cast=cast, gap=gap)
        o = np.lexsort((rec["c"]["childB"], rec["c"]["objB"], rec["c"]["childA"], rec["c"]["objA"], rec["t"]))
        out = np.zeros(len(rec), engine.RECORD_DTYPE)
        out["t"] = rec["t"][o]; out["cctype"] = 1 if cast else 0
        out["idA"] = rec["c"]["objA"][o]; out["childA"] = rec["c"]["childA"][o]; out["idB"] = rec["c"]["objB"][o]; out["childB"] = rec["c"]["childB"][o]
        out["dist"] = rec["c"]["dist"][o]
        parts.append((out, np.asarray(grads, np.float64).reshape(len(rec), 2, rb.K)[o]))
    if not parts:
        return np.zeros(0, engine.RECORD_DTYPE), np.zeros((0, 2, rb.K))
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def _slab_bytes(rec, g):
    return [(int(r["t"]), int(r["cctype"]), int(r["idA"]), int(r["childA"]), int(r["idB"]), int(r["childB"]), float(r["dist"]), gg.tobytes())
            for r, gg in zip(rec, g)]


def _shard_rank_main(rank, world, port, q, gap):
    import torch.distributed as dist
    from oracle import bindings as B
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s, rb, static_T, d, traj = _long_problem()
        sh = multi_gpu.timestep_shard(len(traj), world, rank, gap)
        w = B.oracle_world(s)
        w.set_contact_distance(d)
        rec, g = _window_records(w, rb, static_T, traj, sh, gap)
        pe = w.counter(0)
        w.close()
        rec = multi_gpu.shift_timesteps(rec, sh["lo"])
        recs, gs = multi_gpu.all_gather_slabs(rec, g, dist)          # the one exchange step: every shard's slab on every rank
        merged, mg = multi_gpu.merge_timestep_shards(recs, gs)
        tmax, usum = multi_gpu.reduce_measurement(1.0 + rank, pe, dist)
        q.put((rank, sh, _slab_bytes(merged, mg), usum))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_timestep_shard_covers_every_cost_once():
    for T in (1, 2, 5, 11, 100):
        for world in (1, 2, 3, 8):
            for gap in (1, 2, 3):
                sh = [multi_gpu.timestep_shard(T, world, r, gap) for r in range(world)]
                disc = [t for s_ in sh for t in range(s_["lo"], s_["lo"] + s_["n_discrete"])]
                cast = [t for s_ in sh for t in range(s_["lo"], s_["lo"] + max(s_["stage_hi"] - s_["lo"] - gap, 0))]
                assert disc == list(range(T))                                   # every discrete cost once, in order
                assert cast == list(range(max(T - gap, 0)))                     # every cast pair once
                assert all(s_["stage_hi"] <= T and s_["stage_hi"] - s_["hi"] <= gap for s_ in sh)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gap", [1, 2])
def test_two_gloo_ranks_shard_the_timesteps(gap):
    """ONE trajectory cut by timestep over two ranks: the union of the shards' slabs (t shifted, all-gathered, merged) is the
    single-process slab bit for bit -- contacts, distances, both gradient rows -- and the pair evaluations add up."""
    import torch.multiprocessing as mp
    from oracle import bindings as B
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_rank_main, args=(r, 2, port, q, gap)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda g: g[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s, rb, static_T, d, traj = _long_problem()
    w = B.oracle_world(s)
    w.set_contact_distance(d)
    whole = multi_gpu.timestep_shard(len(traj), 1, 0, gap)
    rec, g = _window_records(w, rb, static_T, traj, whole, gap)
    ref, ref_g = multi_gpu.merge_timestep_shards([rec], [g])
    pe_ref = w.counter(0)
    w.close()
    assert got[0][1]["hi"] == got[1][1]["lo"] and got[0][1]["stage_hi"] == got[0][1]["hi"] + gap    # the halo
    want = _slab_bytes(ref, ref_g)
    assert len(want) > 0 and any(r[1] for r in want) and not all(r[1] for r in want)                # discrete and cast contacts
    for g_ in got:
        assert g_[2] == want                      # every rank holds the whole trajectory's slab, bit for bit, in canonical order
        assert g_[3] == float(pe_ref)             # nothing evaluated twice (no discrete pass on the halo), nothing skipped
