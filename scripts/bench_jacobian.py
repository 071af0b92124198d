import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from geobipy_amd import FdemBatch, synthetic
s = synthetic.syn10_system()
for B, L in [(65536, 8), (8192, 8), (8192, 30), (65536, 3)]:
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=5)
    b = FdemBatch(s, nl, sig, thk, h)
    J = b.sensitivity(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): b.sensitivity(out=J)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/n
    e0.record()
    for _ in range(n): b.forward()
    e1.record(); torch.cuda.synchronize()
    msf = e0.elapsed_time(e1)/n
    print(f"B={B} L={L}: jacobian {ms:.3f} ms -> {B/ms*1e3/1e6:.2f} M/s ; forward {msf:.3f} ms -> {B/msf*1e3/1e6:.2f} M/s ; ratio {ms/msf:.2f}")
