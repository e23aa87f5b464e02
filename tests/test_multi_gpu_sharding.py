"""N > 1 path on CPU (gloo, world_size 2): the arithmetic of the landmark sharding of the global BA and of its one exchange step.

Rank r keeps the residual blocks of the landmarks HOSTED in its keyframe range [r, r + 1) n_kf / world (tsba_shard_of, csrc/tsba_plan.h; the
observations of a frozen landmark go with their target keyframe; all poses replicated), forms its part of the reduced normal equations, and the
parts are all-reduced (sum).  Here the per-rank parts come from the CPU ORACLE (oracle/tsba_oracle.c: the same shard rule restated) and the
collective is torch.distributed over gloo: this checks that the shards of the restatement sum to its unsharded system -- the exchange
arithmetic -- not the product.  The product's own N > 1 path (sharded upload, split kernel sequence, every collective, ring maps, maps with
long-range coupling) runs under `pytest -m gpu` through the in-process communicator: tests/test_gpu_global.py::test_multi_rank_*,
test_device_landmark_shards_sum_to_unsharded_system, tests/test_gpu_far.py::test_two_ranks; over real RCCL in test_multi_gpu_rccl_two_ranks."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from textslam_amd import synth, abi
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = synth.config_global(n_kf=10, n_pt=400, band=5)
    o = abi.options_global()
    o.lm_shard, o.lm_nshard = rank, world
    radius = o.initial_radius
    part = oracle.partial_system(P, o, 0, radius)
    buf = torch.from_numpy(np.concatenate([part["S"].ravel(), part["g"], part["Hd"], [part["cost"]]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)              # the one exchange step of an LM trial
    m = part["g"].size
    S = buf[:m*m].numpy().reshape(m, m).copy()
    g = buf[m*m:m*m + m].numpy().copy()
    Hd = buf[m*m + m:m*m + 2*m].numpy().copy()
    cost = float(buf[-1])
    sc = 1.0/(1.0 + np.sqrt(Hd))
    S[np.diag_indices(m)] += np.clip(sc**2*Hd, o.min_diagonal, o.max_diagonal)/(radius*sc**2)   # pose damping, added once
    if rank == 0:
        o1 = abi.options_global()
        full = oracle.reduced_system(P, o1, 0, radius)
        q.put((float(np.abs(S - full["S"]).max()/np.abs(full["S"]).max()),
               float(np.abs(g - full["g"]).max()/np.abs(full["g"]).max()),
               abs(cost - full["cost"])/full["cost"],
               int(np.array_equal(part["free_idx"], full["free_idx"]))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_reduced_system_allreduce_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dS, dg, dc, same_free = q.get(timeout=10)
    assert dS < 1e-10 and dg < 1e-10 and dc < 1e-12 and same_free == 1


def test_shard_partition_covers_every_block_once(oracle_lib):
    from textslam_amd import synth, abi
    P = synth.config_global(n_kf=8, n_pt=300, band=4)
    o = abi.options_global()
    full = oracle_lib.reduced_system(P, o, 0, 1e4)
    tot = 0.0
    for world in (2, 3, 8):
        tot = 0.0
        Hd = 0.0
        for r in range(world):
            o.lm_shard, o.lm_nshard = r, world
            part = oracle_lib.partial_system(P, o, 0, 1e4)
            tot += part["cost"]; Hd = Hd + part["Hd"]
        assert tot == pytest.approx(full["cost"], rel=1e-12)
        assert np.allclose(Hd, np.diag(full["Hpp"]), rtol=1e-11)
