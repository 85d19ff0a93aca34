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


def window_seeds(base_seed, config_id, windows_per_rank, rank):
    """cfg4: window i of the job uses seed base+config+i; rank r owns i in [r*B, (r+1)*B)."""
    s0 = base_seed + config_id + rank * windows_per_rank
    return [s0 + i for i in range(windows_per_rank)]


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
    t = torch.from_numpy(local).to(device)
    out = [torch.empty_like(t) for _ in range(d.get_world_size())]
    d.all_gather(out, t)
    return np.concatenate([o.cpu().numpy() for o in out], axis=0)
