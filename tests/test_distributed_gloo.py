"""CPU tier: the N > 1 path (sharding rule + summary gather) with world_size 2 and 3 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geobipy_amd.distributed import SummaryGather, shard
        start, n = shard(N)
        # stand-in for the per-sounding summaries a rank's kernel produced: functions of the GLOBAL index
        idx = torch.arange(start, start + n, dtype=torch.float64)
        chi2, logl = idx * 2.0 + 1.0, -idx * 0.5
        g = SummaryGather(N, 2, torch.device("cpu"))
        for _ in range(3):                      # buffers are reused round after round
            work = g.launch(chi2, logl)
            out = g.finish(work)
        if rank == 0:
            ref = torch.arange(N, dtype=torch.float64)
            ok = bool(torch.equal(out[:, 0], ref * 2.0 + 1.0) and torch.equal(out[:, 1], -ref * 0.5))
            ret.put(ok)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 65536), (2, 1001), (3, 10)])
def test_shard_and_gather_gloo(world, N):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_shards_cover_the_batch_exactly():
    from geobipy_amd.distributed import partition, shard
    for N in [0, 1, 7, 8, 65536, 65537]:
        for w in [1, 2, 4, 8]:
            starts, sizes = partition(N, w)
            assert sizes.sum() == N and np.all(np.diff(starts) == sizes[:-1])
            assert sizes.max() - sizes.min() <= 1
            assert [shard(N, r, w) for r in range(w)] == [(int(a), int(b)) for a, b in zip(starts, sizes)]
