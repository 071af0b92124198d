"""The N > 1 exchange executed over RCCL on ONE MI355X.

The builder's boxes have one GPU and the driver's 8-GPU scaling run was skipped in every round so far, so the ``nccl`` side of
geobipy_amd/distributed.py had only ever run over gloo.  RCCL refuses two ranks on one device, but a ONE-rank ``nccl`` process
group is a real communicator: ``init_process_group(device_id=...)``, ``all_gather_into_tensor`` / ``all_reduce`` / ``barrier`` on
device buffers, work handles waited from another stream -- everything the multi-GPU path does except moving bytes over xGMI.
``SummaryGather(force_collective=True)`` / ``gather_rows(force_collective=True)`` make a one-rank group issue the collective a
one-rank job would otherwise skip, and ``bench.py --force-collective`` runs the bench's per-round side-stream gather that way.
What an 8-GPU run can still surprise after this is bandwidth, not correctness.  (Point-to-point ``isend / irecv`` of
``stream_rows_to_root`` needs two ranks -- torch refuses a send to self -- and stays covered by gloo with device-resident
buffers: tests/test_survey.py's two-ranks-on-one-GPU runs.)  Each case runs in its own process: a process group is global state.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ENV = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")

SCRIPT = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from geobipy_amd.distributed import SummaryGather, gather_rows, shard
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
N = 10007
g = SummaryGather(N, 2, dev, force_collective=True)
assert g.collective and g.recv is not None and shard(N) == (0, N)
side = torch.cuda.Stream(device=dev)
rng = np.random.default_rng(5)
for rnd in range(6):                                            # the bench's round: kernel stream -> event -> side stream gather
    a = torch.as_tensor(rng.normal(size=N), device=dev)
    b = torch.as_tensor(rng.normal(size=N), device=dev)
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        work = g.launch(a, b)
    assert work is not None                                      # a collective really was issued
    work.wait()
    out = g.finish()
    torch.cuda.synchronize(dev)
    assert torch.equal(out, torch.stack([a, b], dim=1)), rnd
rows = torch.as_tensor(rng.permutation(N), device=dev)
vals = torch.as_tensor(rng.normal(size=(N, 6)), device=dev)
got = gather_rows(rows, vals, N, force_collective=True)         # the dynamic schedule's exchange: counts, then padded blocks
ref = torch.empty_like(vals); ref[rows] = vals
assert torch.equal(got, ref)
t = torch.tensor([3.5, -1.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                         # the bench's max-over-ranks timing
dist.barrier(); torch.cuda.synchronize(dev)
assert t.tolist() == [3.5, -1.0]
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


@pytest.mark.gpu
def test_summary_gather_and_gather_rows_over_a_one_rank_rccl_group():
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT)], env=dict(ENV, MASTER_PORT="29547"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_bench_rounds_with_the_gather_forced_over_rccl():
    """bench.py's N > 1 round -- fused kernel, event, side stream, all_gather_into_tensor, pending-work wait -- on a one-rank RCCL
    group; the gathered block equals the local summaries and the line keeps the contract's keys."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--soundings", "8192",
                        "--force-collective", "--no-extras", "--no-rjmcmc", "--no-cpu-baseline", "--no-windowed", "--extras-file", ""],
                       env=dict(ENV, MASTER_PORT="29549"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    fc = line["forced_collective"]
    assert fc["backend"] == "nccl" and fc["world"] == 1 and fc["gathered_equals_local"] and fc["rounds"] >= 8
    assert line["n_gpus"] == 1 and line["finite"] and line["value"] > 1e6
