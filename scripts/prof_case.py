"""One workload per invocation, for rocprofv3 (profiles/run_profile_r3.sh): python scripts/prof_case.py <case>
  tdem_config4       BASELINE config 4: 16 384 soundings x 6 layers x 30 gates (tests/golden/config4_30gates.stm), 100 forwards
  jacobian_headline  gbp_fdem_sensitivity on the headline batch (65 536 x 10 frequencies x 8 layers), 100 launches
  config2            BASELINE config 2: 4 096 soundings x 10 frequencies x 5 layers, 400 fused forward + likelihood launches
(the three above warm the clocks for 50 ms exactly as bench.py's per_call does, then time >= 100 launches: a kernel-trace average
 over cold launches read 15 - 17 % above the bench line's ms_per_step in round 3)
  rjmcmc_8192        BASELINE config 5 on one GPU: 8 192 ten-frequency chains, 300 lock-step iterations (reference Jacobian)
  rjmcmc_1024        one GPU's block when config 5 is spread over 8: 1 024 chains, persistent kernel, 2 launches x 1 000 iterations
Prints one line with the measured rate (HIP-synchronised wall time)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):                     # A/B builds of the library (scripts/ab/*.so): measurement only
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, synthetic
case = sys.argv[1]


def per_call(fn, n):
    """bench.py's per_call: ~50 ms of launches to bring the clocks up, then n timed launches (HIP-synchronised wall time)."""
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.05:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

G = os.path.join(ROOT, "tests", "golden")
OPTS = dict(maximum_number_of_layers=int(os.environ.get("GBP_PROF_K", "30")), minimum_depth=1.0, maximum_depth=150.0, initial_relative_error=0.05,
            minimum_relative_error=0.001, maximum_relative_error=0.5, initial_additive_error=5.0, minimum_additive_error=3.0,
            maximum_additive_error=20.0, relative_error_proposal_variance=1e-6, additive_error_proposal_variance=1e-6,
            probability_of_birth=1.0 / 6.0, probability_of_death=1.0 / 6.0, probability_of_perturb=1.0 / 6.0,
            probability_of_no_change=0.5)


def survey(B, L=8):
    system = synthetic.syn10_system()
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=synthetic.SEED + 2)
    clean = FdemBatch(system, nl, sig, thk, h).forward().cpu().numpy()
    return system, nl, sig, thk, h, synthetic.noisy_observations(clean, seed=synthetic.SEED + 3)


if case == "tdem_config4":
    from geobipy_amd.tdem import TdemBatch, TdemSystem
    B, L = 16384, 6
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=synthetic.SEED + 4)
    tb = TdemBatch(TdemSystem(os.path.join(G, "config4_30gates.stm")), nl, sig, thk, h, (-13.0, 0.0, 2.0))
    dt = per_call(tb.forward, 100)
    print(f"CASE tdem_config4: {1e3 * dt:.4f} ms per forward, {B / dt / 1e6:.2f} M evals/s, launches 50 ms warm-up + 100, soundings {B}")
elif case == "jacobian_headline":
    system, nl, sig, thk, h, obs = survey(65536)
    b = FdemBatch(system, nl, sig, thk, h)
    J = b.sensitivity(); torch.cuda.synchronize()
    dt = per_call(lambda: b.sensitivity(out=J), 100)
    print(f"CASE jacobian_headline: {1e3 * dt:.4f} ms per launch, {65536 / dt / 1e6:.2f} M Jacobians/s, launches 50 ms warm-up + 100")
elif case == "config2":
    B, L = 4096, 5
    system = synthetic.syn10_system()
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=synthetic.SEED + 2)
    obs = synthetic.noisy_observations(FdemBatch(system, nl, sig, thk, h).forward().cpu().numpy(), seed=synthetic.SEED + 3)
    bk = [FdemBatch(system, nl, synthetic.redraw_sigma(B, L, seed=synthetic.SEED + 10 + i), thk, h, data=obs, relative_error=np.full(B, 0.05),
                    additive_error=np.full(B, 5.0)) for i in range(4)]
    it = [0]
    def one():
        bk[it[0] % 4].forward_loglike(want_pred=False); it[0] += 1
    dt = per_call(one, 400)
    print(f"CASE config2: {1e3 * dt:.4f} ms per launch, {B / dt / 1e6:.2f} M evals/s, launches 50 ms warm-up + 400, soundings {B}")
elif case in ("rjmcmc_8192", "rjmcmc_1024"):
    B = 8192 if case.endswith("8192") else 1024
    system, nl, sig, thk, h, obs = survey(B)
    dc = DeviceChains(system, h, obs, seed=1, exact_jacobian=False, **OPTS)
    if B == 1024:
        dc.run_mode = 2
    warm, n_it, calls = (50, 300, 1) if B == 8192 else (100, 1000, 2)
    dc.run(warm); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        dc.run(n_it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"CASE {case}: {1e3 * dt / (calls * n_it):.4f} ms per iteration, {B * calls * n_it / dt / 1e6:.2f} M chain-it/s, "
          f"iterations {warm}+{calls * n_it}, chains {B}, mean layers {dc.k.double().mean().item():.2f}")
else:
    raise SystemExit("unknown case " + case)
