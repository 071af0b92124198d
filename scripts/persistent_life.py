"""Where a persistent launch's time goes on bench.py's 1 024-chain block (the first 1 024 soundings of the headline batch, config 5's options):
stage clock of chain 0, mean / longest workgroup life (gbp_rj_debug_stage_ticks), and the chains' layer counts.
python scripts/persistent_life.py [chains 1024] [iterations 2000]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
Btot, L, F = 65536, 8, 10
system = synthetic.syn10_system()
nl, sigma_true, thk, height = synthetic.draw_models(Btot, L, seed=synthetic.SEED + 2)
clean = FdemBatch(system, nl[:B], sigma_true[:B], thk[:B], height[:B]).forward().cpu().numpy()
rngn = np.random.Generator(np.random.PCG64DXSM(synthetic.SEED + 3))
g1 = rngn.normal(size=(Btot, 2 * F)); g2 = rngn.normal(size=(Btot, 2 * F))
obs = clean * (1.0 + 0.05 * g1[:B]) + 5.0 * g2[:B]
opts = dict(maximum_number_of_layers=int(os.environ.get("GBP_K", "30")), minimum_depth=1.0, maximum_depth=150.0, initial_relative_error=0.05,
            minimum_relative_error=0.001, maximum_relative_error=0.5, initial_additive_error=5.0, minimum_additive_error=3.0,
            maximum_additive_error=20.0, relative_error_proposal_variance=1e-6, additive_error_proposal_variance=1e-6,
            probability_of_birth=1.0 / 6.0, probability_of_death=1.0 / 6.0, probability_of_perturb=1.0 / 6.0, probability_of_no_change=0.5)
dc = DeviceChains(system, height[:B], obs, seed=1, exact_jacobian=False, **opts)
dc.run_mode = 2
dc.run(100); torch.cuda.synchronize()
t0 = time.perf_counter(); dc.run(n_it); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B}: {B * n_it / dt / 1e6:.2f} M chain-it/s, {1e6 * dt / n_it:.1f} us per iteration")
tk = (ctypes.c_int64 * 8)()
lib = _lib.load()
kmax = torch.zeros(B, dtype=torch.int32, device=dc.device)
ksum = torch.zeros(B, dtype=torch.float64, device=dc.device)
_lib.check(lib.gbp_rj_debug_stage_ticks(tk, 1))
n_seg, seg = 10, 100
for _ in range(n_seg):
    dc.run(seg)
    kmax = torch.maximum(kmax, dc.k)
    ksum += dc.k.double()
torch.cuda.synchronize()
_lib.check(lib.gbp_rj_debug_stage_ticks(tk, 2))
its = n_seg * seg
print("stages of chain 0, us/it [propose, fm_dlogc_r, newton, fwd|fm_dlogc_p, accept]:", " ".join(f"{tk[i] / max(1, tk[5]) / 100.0:.1f}" for i in range(5)))
print(f"workgroup life per iteration: mean {tk[7] / 100.0 / seg:.1f} us, longest (of any of the {n_seg} launches) {tk[6] / 100.0 / seg:.1f} us")
k_mean = (ksum / n_seg).cpu().numpy(); km = kmax.cpu().numpy()
print("layer count (sampled every %d iterations): mean over chains %.2f; chains whose count reached > 8: %d; largest %d; the ten largest per-chain means: %s" % (
    seg, k_mean.mean(), int((km > 8).sum()), int(km.max()), np.round(np.sort(k_mean)[-10:], 1)))

if os.environ.get("GBP_DEEP_FIRST"):
    # the same block with its deepest chain's sounding in row 0, so that the stage clock (chain 0) shows where a SLOW chain spends its time
    r = int(np.argmax(k_mean))
    perm = np.r_[r, np.delete(np.arange(B), r)]
    dc2 = DeviceChains(system, height[:B][perm], obs[perm], seed=1, exact_jacobian=False, chain_id=perm.astype(np.int64), **opts)
    dc2.run_mode = 2
    dc2.run(100 + n_it); torch.cuda.synchronize()
    _lib.check(lib.gbp_rj_debug_stage_ticks(tk, 1))
    ks = 0.0
    for _ in range(n_seg):
        dc2.run(seg); ks += float(dc2.k[0])
    torch.cuda.synchronize()
    _lib.check(lib.gbp_rj_debug_stage_ticks(tk, 2))
    print("deepest chain in row 0 (mean layers %.1f over the clocked run): stages us/it [propose, fm_dlogc_r, newton, fwd|fm_dlogc_p, accept]: %s; sum %.1f" % (
        ks / n_seg, " ".join(f"{tk[i] / max(1, tk[5]) / 100.0:.1f}" for i in range(5)), sum(tk[i] for i in range(5)) / max(1, tk[5]) / 100.0))
