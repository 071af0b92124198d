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


def _queue_worker(rank, world, port, N, chunk, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from geobipy_amd.distributed import ChunkQueue, gather_rows
        all_counts, ok = [], True
        for job in range(2):                                # two dynamic jobs in one process group: each queue has its own counter
            q = ChunkQueue(N, chunk)
            dist.barrier()                                  # (the ranks of a job start together: spawn / import times differ by seconds)
            rows, vals, mine = [], [], []
            for s, m in q:
                mine.append((s, m))
                idx = torch.arange(s, s + m, dtype=torch.int64)
                rows.append(idx)
                vals.append(torch.stack([idx.double() * 3.0 - 1.0, torch.full((m,), float(rank), dtype=torch.float64)], dim=1))
                time.sleep(0.004 * (1 + 4 * rank))          # uneven chunk times: the slow ranks leave work to the fast one
            r = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.int64)
            v = torch.cat(vals) if vals else torch.zeros((0, 2), dtype=torch.float64)
            out = gather_rows(r, v, N)
            counts = [None] * world
            dist.all_gather_object(counts, len(mine))
            all_counts.append(counts)
            if rank == 0:
                ref = torch.arange(N, dtype=torch.float64) * 3.0 - 1.0
                ok = ok and bool(torch.equal(out[:, 0], ref)) and sum(counts) == q.n_chunks
                owners = out[:, 1].reshape(-1)
            else:
                assert out is None
        if rank == 0:
            ret.put((ok, all_counts, sorted(set(owners.tolist()))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,chunk", [(2, 1000, 64), (3, 130, 16), (4, 2000, 25)])
def test_dynamic_chunk_queue_gloo(world, N, chunk):
    """Every chunk is taken exactly once (atomic counter in the job's store), whichever rank gets to it; rank 0 assembles
    the rows wherever they were computed; a second job in the same process group starts from its own counter (a per-line loop of
    dynamic infer() calls); with uneven chunk times the fast rank ends up with more chunks than the slowest (world 4: 80 chunks)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_queue_worker, args=(r, world, port, N, chunk, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ok, all_counts, owners = ret.get(timeout=5)
    assert ok is True and len(all_counts) == 2
    for counts in all_counts:
        assert sum(counts) == (N + chunk - 1) // chunk
    if (N + chunk - 1) // chunk >= 40:                 # enough chunks for the timing to show: 1 : 5 : 9 : 13 sleep per chunk
        assert all_counts[0][0] > all_counts[0][-1] and all_counts[1][0] > all_counts[1][-1]
    print("chunks per rank:", all_counts, "owners seen:", owners)


def test_chunk_queue_without_a_process_group():
    from geobipy_amd.distributed import ChunkQueue, gather_rows
    assert list(ChunkQueue(10, 4)) == [(0, 4), (4, 4), (8, 2)] and list(ChunkQueue(0, 4)) == []
    a, b = ChunkQueue(10, 4), ChunkQueue(10, 4)
    assert a.key != b.key                               # every queue of a process its own counter key
    rows = torch.tensor([2, 0, 1])
    out = gather_rows(rows, torch.tensor([[2.0], [0.0], [1.0]], dtype=torch.float64), 3)
    assert out[:, 0].tolist() == [0.0, 1.0, 2.0]
    with pytest.raises(RuntimeError):                    # a row missing / twice is an error, never uninitialised memory
        gather_rows(torch.tensor([0, 0, 1]), torch.zeros((3, 1), dtype=torch.float64), 3)


def _stream_worker(rank, world, port, N, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geobipy_amd.distributed import shard, stream_rows_to_root
        start, n = shard(N)
        rows = torch.arange(start, start + n, dtype=torch.int64)
        f = torch.stack([rows.double() * 0.5, rows.double() + 1000.0], dim=1)
        i = (rows[:, None] * 3 + torch.arange(5)[None, :]).to(torch.int32)
        got_rows, got_f, got_i, sizes = [], [], [], []
        for r, (bf, bi) in stream_rows_to_root(rows, [f, i], chunk_rows=7):
            got_rows.append(r); got_f.append(bf); got_i.append(bi); sizes.append(len(r))
        if rank == 0:
            r = np.concatenate(got_rows); order = np.argsort(r)
            ok = (np.array_equal(r[order], np.arange(N)) and np.array_equal(np.concatenate(got_f)[order][:, 1], np.arange(N) + 1000.0)
                  and np.array_equal(np.concatenate(got_i)[order][:, 4], np.arange(N) * 3 + 4) and max(sizes) <= 7
                  and np.concatenate(got_i).dtype == np.int32)
            ret.put(bool(ok))
        else:
            assert not got_rows
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 50), (3, 20), (4, 123)])
def test_stream_rows_to_root_gloo(world, N):
    """Bounded-memory exchange of per-sounding payloads: rank 0 sees every rank's rows in chunks of at most chunk_rows."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, N, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_whole_lines_go_to_the_least_loaded_rank():
    from geobipy_amd.distributed import assign_lines
    assert assign_lines([5, 9, 3, 9, 1], 2) == [[0, 1], [2, 3, 4]]          # 9 | 9, then 5 -> rank 0 (14), 3 and 1 -> rank 1 (13)
    got = assign_lines([5, 9, 3, 9, 1], 2)
    assert sorted(i for r in got for i in r) == [0, 1, 2, 3, 4]
    assert abs(sum([5, 9, 3, 9, 1][i] for i in got[0]) - sum([5, 9, 3, 9, 1][i] for i in got[1])) <= 1
    assert assign_lines([4], 3) == [[0], [], []] and assign_lines([], 2) == [[], []]
    assert assign_lines([7, 7, 7], 3) == [[0], [1], [2]]
