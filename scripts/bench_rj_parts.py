"""Host issue time vs GPU time of the lock-step drivers: python scripts/bench_rj_parts.py [B] [iterations]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):                     # A/B builds of the library (scripts/ab/*.so): this script only, never the product
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, FdemSystem, synthetic
from test_rjmcmc import RESOLVE_OPTIONS
system = synthetic.syn10_system() if os.environ.get("GBP_SYSTEM") == "syn10" else FdemSystem.read(os.path.join(ROOT, "tests", "golden", "resolve.stm"))
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
if os.environ.get("GBP_BIRTH"):                      # deep models: a birth-heavy move mix (the layer count climbs towards maximum_number_of_layers)
    pb = float(os.environ["GBP_BIRTH"])
    o.update(probability_of_birth=pb, probability_of_death=(1.0 - pb) / 3, probability_of_perturb=(1.0 - pb) / 3, probability_of_no_change=(1.0 - pb) / 3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nl, sig, thk, h = synthetic.draw_models(B, 4, seed=synthetic.SEED + 5)
data = synthetic.noisy_observations(FdemBatch(system, nl, sig, thk, h, waves=2).forward().cpu().numpy())
for mode in [int(m) for m in os.environ.get("GBP_MODES", "1,4,0,1,4,0").split(",")]:
    dc = DeviceChains(system, h, data, seed=3, exact_jacobian=False, forward_waves=2, **o)
    dc.run_mode = mode
    dc.run(100); torch.cuda.synchronize()
    t0 = time.perf_counter(); dc.run(n_it); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B} mode {mode}: host issue {1e3 * (t1 - t0) / n_it:.3f} ms/it, total {1e3 * (t2 - t0) / n_it:.3f} ms/it -> {B * n_it / (t2 - t0) / 1e6:.2f} M chain-it/s" + (f", mean layers {dc.t['k'].double().mean().item():.1f}" if os.environ.get("GBP_BIRTH") else ""), flush=True)
