"""N > 1 path on CPU: world_size-2 gloo processes exercise the view sharding,
the lock-step lighting all-reduce and the throughput aggregation that
bench.py uses (no GPU, no compute)."""
import os
import socket

import numpy as np
import pytest


def test_assign_views_partitions_everything():
    from smvs_amd import shard
    for n, w in [(64, 8), (5, 2), (3, 4), (0, 2)]:
        seen = []
        for r in range(w):
            mine = shard.assign_views(n, w, r)
            assert mine == sorted(mine)
            seen += mine
        assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        shard.assign_views(4, 2, 2)
    b = shard.lockstep_batches(5, 2)
    assert b == [[0, 1], [2, 3], [4, None]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from smvs_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        views = shard.assign_views(5, world, rank)
        # per-view lighting systems (deterministic per view id)
        total_A = np.zeros((16, 16)); total_b = np.zeros(16)
        for batch in shard.lockstep_batches(5, world):
            v = batch[rank]
            if v is None:
                A = np.zeros((16, 16)); b = np.zeros(16)
            else:
                rng = np.random.default_rng(100 + v)
                A = rng.random((16, 16)); b = rng.random(16)
            sA, sb = shard.allreduce_lighting(A, b, dist)
            total_A += sA; total_b += sb
        units, secs = shard.aggregate_throughput(10.0 * (rank + 1), 0.5 + rank, dist)
        dist.barrier()
        out.put((rank, views, total_A, total_b, units, secs))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_lighting_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])
    assert results[0][1] == [0, 2, 4] and results[1][1] == [1, 3]
    want_A = np.zeros((16, 16)); want_b = np.zeros(16)
    for v in range(5):
        rng = np.random.default_rng(100 + v)
        want_A += rng.random((16, 16)); want_b += rng.random(16)
    for _, _, A, b, units, secs in results:
        assert np.allclose(A, want_A, rtol=0, atol=1e-12)
        assert np.allclose(b, want_b, rtol=0, atol=1e-12)
        assert units == 30.0 and secs == 1.5   # sum of units, max of time


def test_single_process_paths_are_identity():
    from smvs_amd import shard
    A = np.eye(16); b = np.arange(16.0)
    sA, sb = shard.allreduce_lighting(A, b, None)
    assert np.array_equal(sA, A) and np.array_equal(sb, b)
    assert shard.aggregate_throughput(7, 2.0) == (7.0, 2.0)


def test_bench_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a torch.distributed environment
    re-executes itself under torch.distributed.run with N ranks on 127.0.0.1
    (the driver's own N > 1 command sets RANK and skips this)."""
    import argparse
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        seen["env"] = env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    rc = bench.respawn_under_torchrun(argparse.Namespace(gpus=4))
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_device_pointer_protocol():
    """shard.device_tensor wraps a raw device address through
    __cuda_array_interface__ (no copy): shape, type and address are what torch
    reads."""
    from smvs_amd import shard
    p = shard._DevicePointer(0x7f0000001000, 272)
    cai = p.__cuda_array_interface__
    assert cai["shape"] == (272,) and cai["typestr"] == "<f8"
    assert cai["data"] == (0x7f0000001000, False) and cai["version"] == 2
    A = np.diag(np.arange(1.0, 17.0)); b = np.arange(16.0)
    assert np.allclose(shard.solve_lighting(A, b), b / np.arange(1.0, 17.0))
