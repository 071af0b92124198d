"""Time of the hit-map kernels (csrc/gbp_hitmap.h) against their algorithmic HBM bytes: python scripts/bench_hitmap.py [B]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests")); import hitmap_reference
from geobipy_amd import hitmap
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nv, nz = 250, 440
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
hm = torch.zeros((B, nv, nz), dtype=torch.int32, device=dev)
# layered posteriors: per sounding ~1 200 runs -- 40 (value bin, depth range) bars of random height
for _ in range(40):
    v = torch.randint(0, nv, (B,), device=dev, generator=g)
    lo = torch.randint(0, nz, (B,), device=dev, generator=g); hi = torch.clamp(lo + torch.randint(1, 200, (B,), device=dev, generator=g), max=nz)
    z = torch.arange(nz, device=dev)[None, :]
    bar = ((z >= lo[:, None]) & (z < hi[:, None])).to(torch.int32) * torch.randint(1, 300, (B, 1), device=dev, generator=g, dtype=torch.int32)
    hm[torch.arange(B, device=dev), v] += bar
lmp = torch.zeros(B, dtype=torch.float64, device=dev)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
ms_s, _ = timed(lambda: hitmap.statistics(hm, lmp, 2.3))
ms_r, (ptr, start, val) = timed(lambda: hitmap.runs(hm))
ms_st, _ = timed(lambda: hitmap_reference.statistics_torch(hm[:1024], lmp[:1024], 2.3), 3)
ms_rt, _ = timed(lambda: hitmap_reference.runs_torch(hm[:1024]), 3)
bytes_map = B * nv * nz * 4
runs = int(ptr[-1])
print(f"B={B}: maps {bytes_map / 1e9:.2f} GB, {runs / B:.0f} runs per sounding")
print(f"statistics: {ms_s:.3f} ms -> {(bytes_map + 4 * B * nz * 8) / ms_s / 1e6:.0f} GB/s of algorithmic bytes (one read of the maps + 4 [B, nz] outputs); torch formulation {ms_st * B / 1024:.1f} ms (scaled from 1 024)")
print(f"runs (count + prefix + write): {ms_r:.3f} ms -> {(2 * bytes_map + runs * 8) / ms_r / 1e6:.0f} GB/s (two reads of the maps + 8 B per run); torch formulation {ms_rt * B / 1024:.1f} ms (scaled from 1 024)")
