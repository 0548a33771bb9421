"""Multi-GPU layout of the linearization path: one process per GPU, independent problems (multi-start trajectories,
BASELINE.json configs[4]) partitioned over the ranks.  The data path has NO collective -- a rank's contacts and
gradient rows belong to its own problems; the only exchanges are the barrier and the scalar reductions of the
measurement (max wall time, summed pair evaluations).  Works on any torch.distributed backend: `nccl` (= RCCL over
xGMI) on the GPUs, `gloo` in the CPU tests.
"""
import numpy as np


def partition(n_problems, world_size, rank):
    """Contiguous block of problems for `rank`: sizes differ by at most one, every problem belongs to one rank."""
    if not (0 <= rank < world_size) or n_problems < 0:
        raise ValueError("bad partition request: n=%d world=%d rank=%d" % (n_problems, world_size, rank))
    base, extra = divmod(n_problems, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


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
