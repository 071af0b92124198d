"""Lock-step driver vs persistent per-chain kernel (gbp_rj_run_mode 1 / 2) by block size.
python scripts/bench_rj_modes.py [resolve|syn10] [exact 0|1]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from geobipy_amd import _lib
if os.environ.get("GBP_AB_LIB"):                     # A/B builds of the library (scripts/ab/*.so): this script only, never the product
    _lib.LIB_PATH = os.path.abspath(os.environ["GBP_AB_LIB"])
from geobipy_amd import DeviceChains, FdemBatch, FdemSystem, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "resolve"
exact = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
system = FdemSystem.read(os.path.join(ROOT, "tests", "golden", "resolve.stm")) if which == "resolve" else synthetic.syn10_system()
o = dict(solve_gradient=True, maximum_number_of_layers=30, minimum_depth=1.0, maximum_depth=150.0, minimum_thickness=1.0,
         initial_relative_error=0.05, minimum_relative_error=0.001, maximum_relative_error=0.5, initial_additive_error=5.0,
         minimum_additive_error=3.0, maximum_additive_error=20.0, relative_error_proposal_variance=1e-6,
         additive_error_proposal_variance=1e-6, probability_of_birth=1.0 / 6.0, probability_of_death=1.0 / 6.0,
         probability_of_perturb=1.0 / 6.0, probability_of_no_change=0.5)
for B in [int(v) for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else '256,1024,2048,4096,8192,16384,65536'.split(','))]:
    nl, sig, thk, h = synthetic.draw_models(B, 4, seed=synthetic.SEED + 5)
    clean = FdemBatch(system, nl, sig, thk, h, waves=2).forward().cpu().numpy()
    data = synthetic.noisy_observations(clean)
    out = []
    for mode in (3, 1, 2, 0):
        for fw in ((2,) if mode != 2 else (0, 2)):
            dc = DeviceChains(system, h, data, seed=3, exact_jacobian=exact, forward_waves=fw, **o)
            dc.run_mode = mode
            n_warm, n_it = 200, (2000 if B <= 8192 else 300)
            dc.run(n_warm); torch.cuda.synchronize()
            t0 = time.perf_counter(); dc.run(n_it); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.append(f"mode {mode} fw {fw}: {B * n_it / dt / 1e6:6.2f} M chain-it/s ({dt / n_it * 1e3:.3f} ms/it)")
            if mode == 2 and os.environ.get("GBP_STAGE_CLOCK"):
                import ctypes
                from geobipy_amd import _lib
                tk = (ctypes.c_int64 * 8)()
                _lib.check(_lib.load().gbp_rj_debug_stage_ticks(tk, 1))
                dc.run(200); torch.cuda.synchronize()
                _lib.check(_lib.load().gbp_rj_debug_stage_ticks(tk, 2))
                if tk[5]:
                    out[-1] += " stages us/it [propose, fm_dlogc_r, newton, fwd|fm_dlogc_p, accept]: " + " ".join(f"{tk[i] / tk[5] / 100.0:.1f}" for i in range(5))
                    out[-1] += f"; workgroup life us/it: mean {tk[7] / 100.0 / 200:.1f}, longest {tk[6] / 100.0 / 200:.1f}"
    print(f"{which} exact={int(exact)} B={B:6d}  " + " | ".join(out), flush=True)
