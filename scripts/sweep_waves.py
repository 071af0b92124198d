"""Forward + likelihood rate against batch size and waves per workgroup (the strong-scaling shards of the headline
workload: 65 536 soundings over 1, 2, 4, 8 GPUs = 65 536 ... 8 192 per GPU).  python scripts/sweep_waves.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import FdemBatch, synthetic
from geobipy_amd._lib import load

lib = load()
system = synthetic.syn10_system()
L = 8
for B in (2048, 4096, 8192, 16384, 32768, 65536):
    nl, sg, th, h = synthetic.draw_models(B, L)
    fb = FdemBatch(system, nl, sg, th, h, data=np.full((B, 20), 100.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
    row = []
    for nw in (0, 1, 2, 3, 4, 5, 7, 10):
        fb.waves = nw
        for _ in range(3): fb.forward_loglike(want_pred=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(10, 400000 // B)
        e0.record()
        for _ in range(reps): fb.forward_loglike(want_pred=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row.append(f"nw={nw}: {ms*1e3:7.1f} us {B/ms/1e3:6.2f} M/s")
    print(f"B={B:6d}  " + " | ".join(row), flush=True)
