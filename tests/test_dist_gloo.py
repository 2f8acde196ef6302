"""The N > 1 path on CPU: world_size 2 over gloo.  Frames are sharded, every rank computes its own shard
(here with the oracle standing in for the device kernels) and one all-gather reassembles the outputs."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from molann_amd.dist import all_gather_outputs, shard_bounds, shard_sizes


def test_shard_bounds_cover_all_frames():
    for n in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            ends, total = 0, 0
            for r in range(world):
                a, b = shard_bounds(n, r, world)
                assert a == ends and b >= a
                ends, total = b, total + (b - a)
            assert total == n and max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from build_util import oracle_for_workload, workload_model
        from molann_amd import workloads as wl
        w = wl.get_workload("C3")
        model = workload_model(w)                       # same seed on every rank: replicated weights
        x = w.make_frames(n_frames, seed=5)             # the whole trajectory, for the check only
        a, b = shard_bounds(n_frames, rank, world)
        y_local = oracle_for_workload(w, model, x[a:b], torch.float32)
        y_all = all_gather_outputs(y_local, n_frames)   # the one collective of the path
        y_all2 = all_gather_outputs(y_local)            # sizes discovered by a small all-gather
        full = oracle_for_workload(w, model, x, torch.float32)
        ok = (y_all.shape == full.shape and torch.allclose(y_all, full, atol=1e-6)
              and torch.equal(y_all, y_all2))
        q.put((rank, bool(ok), tuple(y_all.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [256, 257])
def test_two_rank_allgather_matches_single_process(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert all(r[2] == (n_frames, 8) for r in res)
