"""Where a physics workgroup of the sampler spends its life: python scripts/phys_clock.py [B] [iterations]  with GBP_AB_LIB = a
-DGBP_RJ_PHYS_CLOCK build (scripts/build_ab.sh).  Prints, per kind of workgroup (Jacobian at the remapped model / Jacobian at the
proposal / fused forward), the mean time between the stamps in microseconds."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from geobipy_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, FdemSystem, synthetic
from test_rjmcmc import RESOLVE_OPTIONS
system = synthetic.syn10_system()
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 500
nl, sig, thk, h = synthetic.draw_models(B, 4, seed=synthetic.SEED + 5)
data = synthetic.noisy_observations(FdemBatch(system, nl, sig, thk, h, waves=2).forward().cpu().numpy())
dc = DeviceChains(system, h, data, seed=3, exact_jacobian=False, forward_waves=2, **o)
dc.run_mode = int(os.environ.get("GBP_MODE", "0"))
dc.run(200); torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
tk = (ctypes.c_int64 * 64)()
lib.gbp_debug_phys_ticks(tk, 1)
life = (ctypes.c_int64 * 144)()
lib.gbp_debug_phys_life(life, 1)
t0 = time.perf_counter(); dc.run(n_it); torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.gbp_debug_phys_ticks(tk, 0)
print(f"B={B}: {B * n_it / dt / 1e6:.2f} M chain-it/s (clocked build), mean layers {dc.k.double().mean().item():.2f}")
names = {0: ["action", "k_r/height/bins", "math tables + barrier", "t2 + barrier", "passes (thread 0's wave)", "-", "-"],
         8: ["action", "k_r/height/bins", "math tables + barrier", "t2 + barrier", "passes (thread 0's wave)", "-", "-"],
         16: ["action", "k_r/height/bins", "math tables + barrier", "t2/min sigma + barrier", "passes (wave 0)", "barrier (other wave)", "epilogue + loglike"]}
for base, kind in ((0, "stage 0: Jacobian at the remapped model"), (8, "stage 1: Jacobian at the proposal"), (16, "stage 1: fused forward")):
    tot = 0.0
    print(kind)
    for s in range(7):
        n = tk[base + s + 32]
        if n:
            us = tk[base + s] / n / 100.0
            tot += us
            print(f"   {names[base][s]:32s} {us:7.2f} us  (n = {n})")
    print(f"   {'total':32s} {tot:7.2f} us")
lib.gbp_debug_phys_life(life, 0)
print("life of a workgroup (thread 0's wave) by layer count: mean / longest us (workgroups)")
for kind, name in enumerate(("stage 0 Jacobian", "stage 1 Jacobian", "stage 1 forward")):
    row = []
    for L in range(16):
        s, n, m = (life[(kind * 16 + L) * 3 + j] for j in range(3))
        if n:
            row.append(f"L={L}: {s / n / 100.0:.1f} / {m / 100.0:.1f} ({n})")
    print("  ", name, " | ".join(row))
