"""Fused forward / Jacobian rates at the layer counts the sampler lives at (k ~ 3), per waves-per-workgroup choice."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import FdemBatch, FdemSystem, synthetic, _lib
G = os.path.join(ROOT, "tests", "golden")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for name, s in (("resolve(6f)", FdemSystem.read(os.path.join(G, "resolve.stm"))), ("syn10(10f)", synthetic.syn10_system())):
    for L in (1, 3, 8):
        nl, sig, thk, h = synthetic.draw_models(B, L, seed=3, Lmax=30)
        b = FdemBatch(s, nl, sig, thk, h, data=np.full((B, 2 * s.nFrequencies), 100.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
        out = []
        for nw in (1, 2, 4):
            b.waves = nw
            for _ in range(3): b.forward_loglike(want_pred=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): b.forward_loglike(want_pred=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            out.append(f"nw={nw}: {B*s.nFrequencies/dt/1e6:7.1f}M (s,f)/s")
        b.waves = 0
        for _ in range(2): b.fm_dlogc(exact=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): b.sensitivity(exact=True, max_layers=max(L, 8) if L <= 8 else L, bucket=False)
        torch.cuda.synchronize(); dj = (time.perf_counter() - t0) / 10
        print(f"{name} L={L} B={B}: fused forward " + "  ".join(out) + f" | Jacobian {B*s.nFrequencies/dj/1e6:7.1f}M (s,f)/s", flush=True)
