"""Multi-GPU layout of the linearization path: one process per GPU.  Two ways to cut the work, neither with a data-path
collective: independent problems (multi-start trajectories, BASELINE.json configs[4]) partitioned over the ranks, or the
TIMESTEPS of one large problem (configs[3]) block-partitioned with a halo of `gap` poses (timestep_shard /
merge_timestep_shards below).  Computing the linearizations needs NO collective -- a rank's
contacts and gradient rows belong to its own problems; the exchanges of the measurement are the barrier and two scalar
reductions (max wall time, summed pair evaluations).  The one real exchange step of the north-star is optional and
comes AFTER the path: every rank's result slab to every rank (so that a single SQP host can read all of them from one
GPU) with one all-gather of the record counts and one all-gather of fixed-size padded slabs.  On the GPUs
`all_gather_slabs_device` hands the engine's DEVICE slab to RCCL as it lies (tohc_fetch_device: no trip through host
memory); `all_gather_slabs` is the host-memory front end of the same exchange.
Works on any torch.distributed backend: `nccl` (= RCCL over xGMI) on the GPUs, `gloo` in the CPU tests.
"""
import numpy as np


def partition(n_problems, world_size, rank):
    """Contiguous block of problems for `rank`: sizes differ by at most one, every problem belongs to one rank."""
    if not (0 <= rank < world_size) or n_problems < 0:
        raise ValueError("bad partition request: n=%d world=%d rank=%d" % (n_problems, world_size, rank))
    base, extra = divmod(n_problems, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def timestep_shard(T, world_size, rank, gap=1):
    """Rank `rank`'s share of ONE trajectory of T timesteps.  The reference hatches one collision cost per timestep and one
    per (t, t + gap), each reading only its own timesteps' variables (problem_description.cpp:516-529,
    collision_terms.cpp:156-161), so the costs shard by timestep with no exchange: the rank owns the discrete costs of
    [lo, hi) and the cast costs of the pairs starting in [lo, min(hi, T - gap)), and needs the poses [lo, stage_hi) for them,
    stage_hi = min(hi + gap, T) -- its last `gap` poses are the halo (another rank's timesteps).
    -> dict(lo, hi, stage_hi, n_discrete): stage dofs[:, lo:stage_hi], then
    Engine.enqueue_linearization_window(n_discrete, gap=gap) (tohc_enqueue_linearization_window)."""
    if gap < 1:
        raise ValueError("gap must be >= 1")
    lo, hi = partition(T, world_size, rank)
    return dict(lo=lo, hi=hi, stage_hi=min(hi + gap, T), n_discrete=hi - lo)


def shift_timesteps(records, lo):
    """A shard's records count timesteps from its first staged pose; shift them to the trajectory's numbering."""
    out = records.copy()
    out["t"] += lo
    return out


def merge_timestep_shards(recs, grads):
    """The shards' slabs (records already shifted, shift_timesteps; in rank order) as ONE slab in the engine's canonical
    order -- (problem, t, discrete before cast, then the engine's own order inside a shard, which a stable sort keeps) --
    i.e. what a single GPU's tohc_fetch returns for the whole trajectory.  grads: list of [n, 2, K] arrays or Nones."""
    rec = np.concatenate(recs) if recs else np.zeros(0)
    g = None if any(x is None for x in grads) else np.concatenate(grads)
    order = np.lexsort((rec["cctype"] != 0, rec["t"], rec["problem"]))   # stable: ties keep the shard's order
    # (a (problem, t, kind) group lives in exactly one shard, so the order inside it is that shard's canonical order)
    return rec[order], (None if g is None else g[order])


def reduce_measurement(seconds, units, dist=None, device="cpu"):
    """(max over ranks of the wall time, sum over ranks of the work units) -- the bench contract's whole-job rate is
    units_all / seconds_max.  dist: an initialised torch.distributed module, or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds), float(units)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_counts(count, dist=None, device="cpu"):
    """Per-rank record counts on every rank (for a caller that wants to lay the ranks' slabs out in one array)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.array([int(count)], np.int64)
    import torch
    mine = torch.tensor([int(count)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return np.array([int(o.item()) for o in out], np.int64)


def global_problem_ids(records, lo):
    """Records of a rank number their problems from 0; shift them to the job-wide numbering."""
    out = records.copy()
    out["problem"] += lo
    return out


def all_gather_padded(mine, n, dist):
    """The exchange itself, on whatever device `mine` lives on: mine = uint8 tensor [n, row] (this rank's slab, one
    row per record), n = its record count.  One all-gather of the counts, then ONE all-gather of the slabs padded to
    the largest count (fixed-size messages: over xGMI every rank pushes its slab to its 7 peers on 7 distinct links).
    -> (counts int64[world] on the host, uint8 tensor [world, nmax, row] on `mine`'s device: rank r's records are
    out[r, :counts[r]])."""
    import torch
    world = dist.get_world_size()
    counts = gather_counts(n, dist, mine.device)
    nmax = max(int(counts.max()), 1)
    row = mine.shape[1]
    pad = mine
    if mine.shape[0] != nmax:
        pad = torch.zeros((nmax, row), dtype=torch.uint8, device=mine.device)
        if n:
            pad[:n] = mine[:n]                     # device-to-device
    out = torch.empty((world * nmax, row), dtype=torch.uint8, device=mine.device)   # rank-major concatenation
    dist.all_gather_into_tensor(out, pad.contiguous())
    return counts, out.view(world, nmax, row)


class _DeviceBytes:
    """n bytes of device memory owned by someone else (the engine's slab), seen by torch without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}   # torch takes no read-only flag; nobody writes through it


def all_gather_slabs_device(eng, dist, device_index=None):
    """The north-star's exchange step with device buffers in and out: the engine's slab (canonical order, device memory,
    tohc_fetch_device) goes into RCCL as it lies -- no copy to the host and back -- and every rank's records and gradient
    rows end up on every GPU.  -> (counts[world], records uint8 tensor [world, nmax, 152], gradient rows float64 tensor
    [world, nmax, row_doubles] or None), both on this rank's GPU; the SQP host copies out what it needs from ONE GPU."""
    import torch
    d_rec, d_g, n, rowd = eng.fetch_device()
    dev = torch.device("cuda", torch.cuda.current_device() if device_index is None else device_index)
    rec = (torch.as_tensor(_DeviceBytes(d_rec, n * 152), device=dev).view(n, 152) if n
           else torch.zeros((0, 152), dtype=torch.uint8, device=dev))
    counts, recs = all_gather_padded(rec, n, dist)
    grads = None
    if rowd:
        g = (torch.as_tensor(_DeviceBytes(d_g, n * rowd * 8), device=dev).view(n, rowd * 8) if n
             else torch.zeros((0, rowd * 8), dtype=torch.uint8, device=dev))
        _, gs = all_gather_padded(g, n, dist)
        grads = gs.view(torch.float64).view(gs.shape[0], gs.shape[1], rowd)
    return counts, recs, grads


def all_gather_slabs(records, grads, dist=None, device="cpu"):
    """Host-memory front end of the same exchange (the CPU tests run it over gloo): every rank's (records, grads) on
    every rank, in rank order, problems renumbered job-wide by the caller.  records: numpy structured array [n]
    (engine.RECORD_DTYPE, 152 B); grads: float64 [n, 2, K] or None.  Returns (list of records per rank, list of grads
    per rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [records], [grads]
    import torch
    world = dist.get_world_size()
    n = len(records)
    rec_b = records.dtype.itemsize
    g_b = 0 if grads is None else int(np.prod(grads.shape[1:])) * 8
    buf = np.zeros((n, rec_b + g_b), np.uint8)
    if n:
        buf[:, :rec_b] = np.ascontiguousarray(records).view(np.uint8).reshape(n, rec_b)
        if g_b:
            buf[:, rec_b:] = np.ascontiguousarray(grads, np.float64).view(np.uint8).reshape(n, g_b)
    counts, out = all_gather_padded(torch.from_numpy(buf).to(device), n, dist)
    host = out.cpu().numpy()
    recs, gs = [], []
    for r in range(world):
        c = int(counts[r])
        recs.append(np.ascontiguousarray(host[r, :c, :rec_b]).view(records.dtype).reshape(c))
        gs.append(None if not g_b else np.ascontiguousarray(host[r, :c, rec_b:]).view(np.float64).reshape((c,) + tuple(grads.shape[1:])))
    return recs, gs
