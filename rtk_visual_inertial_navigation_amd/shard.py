"""Sharding of independent windows over ranks (one process per GPU).

The path has no exchange step: windows are independent units, so ranks take
disjoint, contiguous blocks of window seeds and never communicate inside a solve.  The only
collectives are harness-side: max of the wall time, sum of iteration counts, and a gather of the
per-window {final_cost, iterations, termination} records.  Backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def partition(n_windows, world, rank):
    """Static block distribution of the job's windows over the ranks (512 windows, 64 per GPU at 8):
    (first window, count) of `rank`; the first n_windows % world ranks take one more."""
    q, r = divmod(n_windows, world)
    return rank * q + min(rank, r), q + (1 if rank < r else 0)


def window_seeds(base_seed, config_id, count, rank=0, first=None):
    """cfg4: window i of the job uses seed base+config+i; this rank owns `count` windows starting at `first`
    (default: rank * count, the weak-scaling layout)."""
    s0 = base_seed + config_id + (rank * count if first is None else first)
    return [s0 + i for i in range(count)]


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def _tensor(vals, device):
    import torch
    return torch.tensor(vals, dtype=torch.float64, device=device)


def allreduce(values, op="max", device="cpu"):
    """values: list of floats -> list reduced over ranks (identity when not distributed)."""
    d = _dist()
    if d is None:
        return list(values)
    t = _tensor(list(values), device)
    d.all_reduce(t, op=d.ReduceOp.MAX if op == "max" else d.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]


def gather_summaries(local, device="cpu"):
    """local: float64 array [B][3] = final_cost, iterations, termination per window of this rank.
    Returns the [world*B][3] array on every rank, rank-major (window i of the job at row i)."""
    d = _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    if d is None:
        return local
    import torch
    # ranks may own different numbers of windows (strong partition of a job that does not divide evenly): gather the counts, pad to the
    # largest, trim after the gather
    world = d.get_world_size()
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    d.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    width = local.shape[1] if local.ndim == 2 else 3
    pad = np.zeros((max(counts), width)); pad[:local.shape[0]] = local.reshape(-1, width)
    t = torch.from_numpy(pad).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    d.all_gather(out, t)
    return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0)
