"""Sharding soundings over the GPUs of a node and gathering per-sounding summaries.

Soundings are independent (the reference farms them one per MPI rank,
inversion/Inference3D.py:518-635), so the only exchange is ONE gather of the per-sounding
summaries (chi^2, logL: 16 B per sounding), assembled on rank 0.  One process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  Blocks follow the reference's
own rule ``loadBalance1D_shrinkingArrays`` (base/MPI.py:172-201).
"""
import numpy as np
import torch
import torch.distributed as dist


def partition(N, nChunks):
    """starts[nChunks], sizes[nChunks]: equal blocks, the first N % nChunks blocks get one extra."""
    sizes = np.full(nChunks, N // nChunks, dtype=np.int64)
    sizes[: N % nChunks] += 1
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    return starts, sizes


def shard(N, rank=None, world=None):
    """(start, size) of this rank's contiguous block of the sounding index."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    starts, sizes = partition(N, world)
    return int(starts[rank]), int(sizes[rank])


class SummaryGather:
    """Gathers [rows_local, C] fp64 summaries of every rank into one [N, C] tensor on rank 0 (one
    ``all_gather_into_tensor`` per round).

    Blocks may differ by one row, so every rank contributes a block padded to the largest size and
    rank 0 strips the padding; buffers are allocated once.  ``launch`` enqueues the collective on the
    current stream (or the given side stream, so that it overlaps the next round's kernel) and returns
    without synchronising.
    """

    def __init__(self, N, C, device, group=None, force_collective=False):
        """``force_collective``: issue the collective even in a one-rank group (a one-rank job needs none and skips it by
        default) -- how the RCCL path (communicator set-up, stream semantics, ``all_gather_into_tensor`` on device buffers) is
        exercised on a box with ONE GPU (tests/test_rccl_single_gpu.py)."""
        self.N, self.C, self.group = N, C, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        self.starts, self.sizes = partition(N, self.world)
        self.pad = int(self.sizes.max())
        self.send = torch.zeros((self.pad, C), dtype=torch.float64, device=device)
        # all_gather_into_tensor is the one collective every backend implements natively (RCCL ring /
        # direct all-gather over xGMI); the 16 B/sounding payload makes the extra copies on ranks != 0 free
        self.recv = torch.empty((self.world * self.pad, C), dtype=torch.float64, device=device) \
            if self.collective else None
        self.out = torch.empty((N, C), dtype=torch.float64, device=device) if self.rank == 0 else None

    def launch(self, *columns):
        """columns: C tensors of shape [rows_local]; copies them into the send block and starts the gather."""
        n = int(self.sizes[self.rank])
        for c, col in enumerate(columns):
            self.send[:n, c].copy_(col[:n])
        if not self.collective:
            self.out[:n].copy_(self.send[:n])
            return None
        return dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)

    def finish(self, work=None):
        """Wait for the collective and assemble the [N, C] result on rank 0 (None elsewhere)."""
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        if self.collective:
            for r in range(self.world):
                s, n = int(self.starts[r]), int(self.sizes[r])
                self.out[s:s + n].copy_(self.recv[r * self.pad: r * self.pad + n])
        return self.out


class ChunkQueue:
    """Dynamic scheduling of chunks of soundings over the ranks of a job.

    The reference's master hands the next data point to whichever worker reports back (Inference3D.py:518-635), because
    soundings take different numbers of iterations to burn in and a static split leaves ranks idle.  Here there is no master
    rank and there are no messages: every rank draws the index of its next chunk from ONE atomic counter in the job's
    key-value store (``TCPStore.add``, the store ``init_process_group`` already created), until the chunks run out.  A chunk
    is a contiguous block of soundings, large enough to fill a GPU (the unit of work of the device sampler is a block of
    chains, not one sounding).  Iterating yields (start, size); without an initialised process group it runs through all chunks.
    """

    _jobs = 0          # queues made so far in this process: every rank makes them in the same order, so the n-th queue of a job is
                       # the n-th on every rank and gets its own counter in the store (a second dynamic infer() -- a per-line loop --
                       # must not find the first one's counter already past its chunks)

    def __init__(self, n_items, chunk, key=None, store=None):
        assert chunk >= 1
        self.n_items, self.chunk = int(n_items), int(chunk)
        self.n_chunks = (self.n_items + self.chunk - 1) // self.chunk
        if key is None:
            key = "gbp_chunk_queue/{}".format(ChunkQueue._jobs)
            ChunkQueue._jobs += 1
        self.key = key
        self._local = 0
        self.store = store
        if store is None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self.store = distributed_c10d._get_default_store()

    def take(self):
        """Index of the next chunk, or None when all are taken."""
        if self.store is None:
            c, self._local = self._local, self._local + 1
        else:
            c = int(self.store.add(self.key, 1)) - 1
        return c if c < self.n_chunks else None

    def __iter__(self):
        while True:
            c = self.take()
            if c is None:
                return
            s = c * self.chunk
            yield s, min(self.chunk, self.n_items - s)


def gather_rows(rows, values, N, group=None, force_collective=False):
    """[N, C] on rank 0 (None elsewhere) from every rank's (rows[m] int64, values[m, C] float64): the exchange at the end of a
    dynamically scheduled job, where a rank's rows are not a contiguous block.  Two collectives: the row counts, then one
    ``all_gather_into_tensor`` of blocks padded to the largest count, the row index travelling as an extra column."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev, C = values.device, values.shape[1]
    if world == 1 and not (force_collective and dist.is_initialized()):      # (force_collective: see SummaryGather)
        if int(rows.numel()) != N or int(torch.unique(rows).numel()) != N:
            raise RuntimeError("gather_rows: every row must arrive exactly once")
        out = torch.full((N, C), float("nan"), dtype=torch.float64, device=dev)
        out[rows] = values
        return out
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([rows.numel()], dtype=torch.int64, device=dev), group=group)
    pad = int(counts.max())
    send = torch.zeros((max(pad, 1), C + 1), dtype=torch.float64, device=dev)
    send[:rows.numel(), 0] = rows.to(torch.float64)
    send[:rows.numel(), 1:] = values
    recv = torch.empty((world * max(pad, 1), C + 1), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    if rank != 0:
        return None
    # every row exactly once: a queue that handed out nothing (or a chunk twice) must not come back as uninitialised memory
    if int(counts.sum()) != N:
        raise RuntimeError("gather_rows: {} rows arrived for {} soundings".format(int(counts.sum()), N))
    out = torch.full((N, C), float("nan"), dtype=torch.float64, device=dev)
    seen = torch.zeros(N, dtype=torch.bool, device=dev)
    for r in range(world):
        n = int(counts[r])
        block = recv[r * max(pad, 1): r * max(pad, 1) + n]
        idx = block[:, 0].to(torch.int64)
        out[idx] = block[:, 1:]
        seen[idx] = True
    if not bool(seen.all()):
        raise RuntimeError("gather_rows: some soundings never arrived (and others twice)")
    return out


def assign_lines(counts, world):
    """Whole flight lines to ranks (the "lines" schedule: a rank that owns a line writes that line's results file itself, so no
    posterior row travels): the longest line first, each to the rank with the fewest soundings so far (ties: the lowest rank).
    ``counts``: soundings per line -> one ascending list of line indices per rank; deterministic, the same on every rank."""
    load, out = [0] * world, [[] for _ in range(world)]
    for i in sorted(range(len(counts)), key=lambda j: (-int(counts[j]), j)):
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(i)
        load[r] += int(counts[i])
    return [sorted(x) for x in out]


def stream_rows_to_root(rows, blocks, chunk_rows=64, group=None):
    """Generator for rank 0: (rows int64[m] numpy, [block[m, W_i] numpy ...]) chunk by chunk, over every rank's rows -- the
    bounded-memory exchange for per-sounding payloads too large to gather at once (config 5's conductivity-depth hit maps are
    440 KB per sounding: SURVEY 8e).  ``rows``: this rank's global row indices (int64 tensor); ``blocks``: tensors [len(rows),
    W_i] on one device, any dtypes.  Point-to-point: rank r sends its chunks in order, rank 0 receives into one reusable buffer
    per block; other ranks get an empty generator (iterate it all the same -- that is what sends)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = blocks[0].device
    m = int(rows.numel())
    own = lambda x: (x.clone() if x.device.type == "cpu" else x.cpu()).numpy()        # (a receive buffer is reused: hand out copies)
    to_host = lambda r, bs, a, b: (own(r[a:b]), [own(x[a:b]) for x in bs])
    # this rank's own rows are nobody's buffer: views of host tensors as they are (the hit maps are 440 KB per row -- a copy of every
    # chunk was a fifth of a survey's wall time), one device -> host copy otherwise
    mine = lambda x: (x if x.device.type == "cpu" else x.cpu()).numpy()
    own_rows = lambda a, b: (mine(rows[a:b]), [mine(x[a:b]) for x in blocks])
    if world == 1:
        for a in range(0, m, chunk_rows):
            yield own_rows(a, min(m, a + chunk_rows))
        return
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([m], dtype=torch.int64, device=dev), group=group)
    counts = counts.cpu().tolist()
    if dist.get_backend(group) == "gloo" and dev.type != "cpu":      # gloo's point-to-point wants host memory: stage the chunks
        dev = torch.device("cpu")
        stage = lambda x: x.cpu()
    else:
        stage = lambda x: x.contiguous()
    # Rounds: in round i every peer that still has a chunk i sends it, and rank 0 has receives posted for ALL of them at once
    # (one reusable buffer set per peer: (world - 1) x chunk_rows rows of memory), so no rank waits for another rank to be
    # drained completely -- at config 5's 3.5 GB of hit maps per GPU the peers' sends overlap instead of queueing behind rank 1.
    n_rounds = max((c_ + chunk_rows - 1) // chunk_rows for c_ in counts) if counts else 0
    if rank == 0:
        for a in range(0, m, chunk_rows):
            yield own_rows(a, min(m, a + chunk_rows))
        peers = [r for r in range(1, world) if counts[r] > 0]
        rbuf = {r: torch.empty(chunk_rows, dtype=torch.int64, device=dev) for r in peers}
        bufs = {r: [torch.empty((chunk_rows,) + tuple(b.shape[1:]), dtype=b.dtype, device=dev) for b in blocks] for r in peers}
        for i in range(n_rounds):
            posted = []
            for r in peers:
                a = i * chunk_rows
                if a >= counts[r]:
                    continue
                n = min(chunk_rows, counts[r] - a)
                works = [dist.irecv(rbuf[r][:n], src=r, group=group)] + [dist.irecv(buf[:n], src=r, group=group) for buf in bufs[r]]
                posted.append((r, n, works))
            for r, n, works in posted:
                for w in works:
                    w.wait()
                yield to_host(rbuf[r], bufs[r], 0, n)
    else:
        for a in range(0, m, chunk_rows):
            b = min(m, a + chunk_rows)
            works = [dist.isend(stage(rows[a:b]), dst=0, group=group)] + [dist.isend(stage(x[a:b]), dst=0, group=group) for x in blocks]
            for w in works:
                w.wait()
