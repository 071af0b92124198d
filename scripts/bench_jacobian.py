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

# rjMCMC-like ragged population: 90 % of the models have 1-6 layers, 10 % have 7-30
rng = np.random.default_rng(0)
B, Lmax = 65536, 30
nl = np.where(rng.uniform(size=B) < 0.9, rng.integers(1, 7, size=B), rng.integers(7, Lmax + 1, size=B)).astype(np.int32)
_, sig, thk, h = synthetic.draw_models(B, Lmax, seed=6)
b = FdemBatch(s, nl, sig, thk, h)
for bucket in (False, True):
    J = b.sensitivity(bucket=bucket); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): b.sensitivity(out=J, bucket=bucket)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/5
    print(f"ragged B={B} (mean L {nl.mean():.1f}, max 30) bucket={bucket}: jacobian {ms:.3f} ms -> {B/ms*1e3/1e6:.2f} M/s")
