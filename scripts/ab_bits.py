"""Are two builds of the library bit-identical, and how fast are their forward / Jacobian kernels?
    python scripts/ab_bits.py dump OUT.npz      (GBP_AB_LIB = the library to load, unset = the product's)
    python scripts/ab_bits.py compare A.npz B.npz
dump: headline batch (65 536 x 10 frequencies x 8 layers) forward + likelihood with the default abscissa windows and with all
abscissae, the Jacobian of 8 192 soundings, a ragged 30-layer batch, and the state of 2 048 chains after 300 iterations; prints the
kernels' rates (HIP-synchronised wall time, clocks warmed as bench.py does)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        if not same:
            bad += 1
            d = np.nanmax(np.abs(a[k].astype(float) - b[k].astype(float))) if a[k].shape == b[k].shape else np.nan
            print(f"DIFFERENT {k}: max abs difference {d:.3e}")
    print("BIT-IDENTICAL" if bad == 0 else f"{bad} arrays differ", f"({len(a.files)} arrays)")
    raise SystemExit(1 if bad else 0)

import torch
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, synthetic
from test_rjmcmc import RESOLVE_OPTIONS


def per_call(fn, n):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {}
system = synthetic.syn10_system()
B = 65536
nl, sig, thk, h = synthetic.draw_models(B, 8, seed=synthetic.SEED + 2)
fb = FdemBatch(system, nl, sig, thk, h)
clean = fb.forward().cpu().numpy()
obs = synthetic.noisy_observations(clean, seed=synthetic.SEED + 3)
for name, kw in (("window", {}), ("all", dict(hankel_eps_ppm=0.0))):
    b = FdemBatch(system, nl, sig, thk, h, data=obs, relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0), **kw)
    r = b.forward_loglike()
    out["pred_" + name] = b.predicted.cpu().numpy(); out["chi2_" + name] = b.chi2.cpu().numpy(); out["logL_" + name] = b.logL.cpu().numpy()
    dt = per_call(lambda: b.forward_loglike(want_pred=False), 200)
    print(f"forward + likelihood, {name}: {1e3 * dt:.4f} ms per launch, {B / dt / 1e6:.2f} M evals/s")
bj = FdemBatch(system, nl[:8192], sig[:8192], thk[:8192], h[:8192])
J = bj.sensitivity()
out["J"] = J.cpu().numpy()
dt = per_call(lambda: bj.sensitivity(out=J), 200)
print(f"Jacobian 8192: {1e3 * dt:.4f} ms per launch, {8192 / dt / 1e6:.2f} M Jacobians/s")
nl2, sig2, thk2, h2 = synthetic.draw_models(4096, 30, seed=synthetic.SEED + 7)
b2 = FdemBatch(system, nl2, sig2, thk2, h2)
out["pred_ragged30"] = b2.forward().cpu().numpy()
out["J_ragged30"] = b2.sensitivity().cpu().numpy()
out["fm_J"] = b2.fm_dlogc().cpu().numpy(); out["fm_pred"] = b2.predicted.cpu().numpy()
o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
Bc = 2048
for mode in (0, 2):
    dc = DeviceChains(system, h[:Bc], obs[:Bc], seed=3, exact_jacobian=False, **o)
    if mode:
        dc.run_mode = mode
    dc.run(300); torch.cuda.synchronize()
    for k in ("k", "sigma", "edges", "like", "pred", "J", "rel", "add"):
        out[f"chains{mode}_{k}"] = dc.t[k].cpu().numpy()
np.savez(sys.argv[2], **out)
print("wrote", sys.argv[2], len(out), "arrays")
