"""world_size-2 gloo test of the N>1 path on CPU: disjoint window shards, harness collectives,
and that the gathered per-window records equal what one process computes for the whole job.
(The per-window solves here run through the CPU oracle — test infrastructure — because there is
no GPU in this container; on the GPU box the same sharding code drives libswf_hip.so.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from rtk_visual_inertial_navigation_amd import synth, shard
    from rtk_visual_inertial_navigation_amd.flat import default_options
    import oracle_binding as ob
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = shard.window_seeds(synth.BASE_SEED, 4, B, rank)
    recs = []
    for s in seeds:
        w = synth.make_window(3, K=4, F=8, S=5, seed=s)
        sm, _ = ob.solve(w, default_options(max_num_iterations=4), export=False)
        recs.append([sm.final_cost, sm.num_iterations, sm.termination])
    allr = shard.gather_summaries(np.array(recs))
    tmax = shard.allreduce([1.0 + rank], "max")[0]
    isum = shard.allreduce([sum(r[1] for r in recs)], "sum")[0]
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([allr.ravel(), [tmax, isum], seeds]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    import torch.multiprocessing as mp
    world, B = 2, 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rtk_visual_inertial_navigation_amd import synth, shard
    from rtk_visual_inertial_navigation_amd.flat import default_options
    import oracle_binding as ob
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npy")); r1 = np.load(os.path.join(str(tmp_path), "rank1.npy"))
    n = world * B * 3
    assert np.array_equal(r0[:n + 2], r1[:n + 2])                     # every rank sees the same gathered job
    s0, s1 = r0[n + 2:], r1[n + 2:]
    assert len(set(s0) | set(s1)) == world * B and not (set(s0) & set(s1))   # disjoint shards
    assert r0[n] == 2.0                                               # max over ranks of (1 + rank)
    # single-process reference for the whole job, same seeds in job order
    exp = []
    for rank in range(world):
        for s in shard.window_seeds(synth.BASE_SEED, 4, B, rank):
            w = synth.make_window(3, K=4, F=8, S=5, seed=s)
            sm, _ = ob.solve(w, default_options(max_num_iterations=4), export=False)
            exp.append([sm.final_cost, sm.num_iterations, sm.termination])
    exp = np.array(exp)
    assert np.array_equal(r0[:n].reshape(-1, 3), exp)                  # bit-identical: windows are independent
    assert r0[n + 1] == exp[:, 1].sum()
