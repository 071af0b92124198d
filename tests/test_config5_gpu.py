"""BASELINE config 5 AT SIZE on one GPU: 8 192 soundings x 10 000 rjMCMC iterations (the reference's resolve_options: birth /
death / perturb / none = 1/6, 1/6, 1/6, 1/2, up to 30 layers, reference-expression Jacobian = the pinned parity mode).

  (i)   64 of the 8 192 chains are replayed for the FULL 10 000 iterations on the CPU -- host emulation of the stage kernels
        (tests/rj_emul.py, built on rjmcmc.py which reproduces the reference's own seeded chains) around the C oracle's
        forward and Jacobian, same counter-based random streams -- and compared at 100 checkpoints (layer count, accepted
        steps, misfit) and in their final layer-count / interface-depth histograms: exact-match count and first divergence
        are printed ("posterior histograms bit-matching CPU seeds").  Measured: 58 of 64 chains
        identical over all 10 000 iterations, the other 6 part between iterations 3 700 and 9 500.  WHY (round 3,
        scripts/replay_arms.py -> profiles/r3/replay_arms_gpu_vs_cpu.json, scripts/replay_sensitivity.py ->
        profiles/r3/replay_sensitivity_cpu.json): the same replay in four arms -- {reference Jacobian expression, exact
        derivative} x {per-sounding abscissa window, all 120 abscissae} -- gives 58 / 58 / 59 / 59 of 64, so neither the
        reference's non-derivative Jacobian (round 2's guess) nor the window is the cause; the chains that match carry a
        relative misfit difference of 5e-11 (median) that does NOT grow over the 10 000 iterations.  And the CPU chain run
        twice, once with the oracle's outputs perturbed at the level two correct implementations differ by (prediction +-
        3e-9 ppm, Jacobian 1e-10 relative), parts from itself at the same rate: 8 of 48 chains in 10 000 iterations.  In the
        ~50 iterations before such a split the log acceptance ratios of the two runs differ by up to O(1) where the
        stochastic-Newton precision J'PJ + Wm'Wm is ill-conditioned (cond 1e4 ... 4e5): the chain map itself is expansive
        there, whatever computes the forward.  SURVEY 7-5's estimate (one split per 1e8 iterations) assumed a
        well-conditioned map; the measured rate, device-vs-CPU and CPU-vs-CPU alike, is ~1e-5 per iteration.  Required:
        every chain identical for the first 2 000 iterations, >= 56 of 64 to the end (the arms measured 58 - 59), and the matching chains within 5e-9
        (median) / 5e-3 (any checkpoint: the stretches above) of the CPU misfit.
  (ii)  invariants on all 8 192 chains: finite state, structural constraints, posterior counts, cached prediction / misfit /
        likelihood equal to a from-scratch evaluation.
  (iii) the same survey run as two blocks of 4 096 (what two GPUs would do) ends bit-identical, row for row.
"""
import os
import time

import numpy as np
import pytest

import config5_replay
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

S, N_IT, EVERY, N_REPLAY = 8192, 10000, 100, 64


def _survey():
    """Synthetic Resolve survey: 4-layer earths (SURVEY 8d generator), 5 % + 5 ppm noise."""
    from geobipy_amd import FdemBatch, FdemSystem, synthetic
    system = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    nl, sigma, thk, height = synthetic.draw_models(S, 4, seed=synthetic.SEED + 5)
    clean = FdemBatch(system, nl, sigma, thk, height, waves=2).forward().cpu().numpy()
    noise = np.random.Generator(np.random.PCG64DXSM(synthetic.SEED + 6)).normal(size=clean.shape)
    data = clean + noise * np.sqrt((0.05 * clean) ** 2 + 5.0 ** 2)
    return system, height, data


def _chains(system, height, data, first=0):
    from geobipy_amd import DeviceChains
    from test_rjmcmc import RESOLVE_OPTIONS
    o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
    return DeviceChains(system, height, data, seed=20260928, exact_jacobian=False, first_chain=first, forward_waves=2, **o)


def test_config5_at_size_matches_cpu_replays_and_is_shard_independent():
    from geobipy_amd import FdemBatch
    from geobipy_amd import rjmcmc_gpu as rg
    system, height, data = _survey()
    dc = _chains(system, height, data)
    rows = np.linspace(0, S - 1, N_REPLAY).astype(int)
    specs = config5_replay.specs_from_device(dc, rows, "resolve", N_IT, EVERY, data, height)
    pool, pending = config5_replay.start(specs)                 # CPU replays run while the GPU works
    try:
        t0 = time.perf_counter()
        marks = []
        rows_t = torch.as_tensor(rows, device=dc.device)
        for _ in range(N_IT // EVERY):
            dc.run(EVERY)
            marks.append(torch.stack([dc.k[rows_t].double(), dc.n_accepted[rows_t].double(), dc.misfit[rows_t]], dim=1).cpu().numpy())
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        marks = np.array(marks)                                   # [checkpoints, N_REPLAY, 3]
        # (ii) invariants on all chains
        k = dc.k.cpu().numpy()
        assert k.min() >= 1 and k.max() <= dc.K and dc.iteration == N_IT
        for n in ("sigma", "rel", "add", "pred", "misfit", "like", "prior", "best_sigma"):
            assert bool(torch.isfinite(getattr(dc, n)).all()), n
        thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64))
        assert bool((torch.where(thk > 0, thk, torch.full_like(thk, 9.0)) > dc.min_width).all())
        assert bool((dc.k_hist.sum(dim=1) == N_IT).all()) and bool((dc.n_accepted > 0).all())
        fb = FdemBatch(system, k, dc.sigma.cpu().numpy(), thk.cpu().numpy(), height, data=data,
                       relative_error=dc.rel[:, 0].cpu().numpy(), additive_error=dc.add[:, 0].cpu().numpy(), waves=2)
        chi2, logl = fb.forward_loglike()
        assert torch.allclose(fb.predicted, dc.pred, rtol=1e-9, atol=1e-7) and torch.allclose(chi2, dc.misfit, rtol=1e-7)
        assert torch.allclose(logl, dc.like, rtol=1e-8)
        # (iii) two blocks of 4 096 = the one block of 8 192, bit for bit
        t0 = time.perf_counter()
        halves = [_chains(system, height[s], data[s], first=s.start).run(N_IT) for s in (slice(0, S // 2), slice(S // 2, S))]
        torch.cuda.synchronize()
        t_halves = time.perf_counter() - t0
        for n in ("k", "edges", "sigma", "rel", "add", "misfit", "n_accepted", "k_hist", "edge_hist", "rel_hist", "add_hist", "best_sigma"):
            assert torch.equal(torch.cat([getattr(h, n) for h in halves]), getattr(dc, n)), n
        # (i) the CPU replays
        t0 = time.perf_counter()
        results = pending.get(timeout=900)
        t_wait = time.perf_counter() - t0
    finally:
        pool.terminate()
    cmp = config5_replay.compare(results, marks, dc.k_hist[rows_t].cpu().numpy(), dc.edge_hist[rows_t].cpu().numpy(), rows)
    exact = [c for c in cmp if c["first_divergent_checkpoint"] < 0 and c["histograms_equal"]]
    diverged = [(c["row"], (c["first_divergent_checkpoint"] + 1) * EVERY) for c in cmp if c["first_divergent_checkpoint"] >= 0]
    print(f"config 5 at size: {S} chains x {N_IT} iterations in {t_gpu:.1f} s on one GPU ({S * N_IT / t_gpu / 1e6:.1f} M chain-it/s, "
          f"checkpointed every {EVERY}); two blocks of {S // 2}: {t_halves:.1f} s; CPU replay of {N_REPLAY} chains: waited {t_wait:.1f} s more; "
          f"exact matches {len(exact)}/{N_REPLAY}, diverged (row, by iteration): {diverged}; "
          f"rel misfit difference of the matching chains: median {np.median([c['max_rel_misfit_diff'] for c in exact]):.1e}, "
          f"max {max([c['max_rel_misfit_diff'] for c in exact], default=0):.1e}")
    # bars = what the four replay arms measured (profiles/r3/replay_arms_gpu_vs_cpu.json: 58 / 58 / 59 / 59 of 64 identical to the
    # end, median misfit difference of the matching chains 5e-11), with a margin of two chains: a real regression cannot hide
    assert len(exact) >= 56, cmp
    assert all(c["first_divergent_checkpoint"] < 0 or (c["first_divergent_checkpoint"] + 1) * EVERY > 2000 for c in cmp), cmp
    assert all(c["max_rel_misfit_diff"] < 5e-3 for c in exact)
    # (median over the matching chains of their LARGEST difference at any checkpoint: 2.8e-9 measured, deterministic; the 5e-11 of the
    #  arms' table is the median of the differences at the END of the run)
    assert np.median([c["max_rel_misfit_diff"] for c in exact]) < 5e-9
