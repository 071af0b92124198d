"""Device-resident rjMCMC on time-domain data (geobipy_amd.tdem.TdemDeviceChains -> gbp_rj_run_td), one system.

No reference pin exists for the TDEM path (gatdaem1d is absent); what is checked here is that the sampler
around it is the FDEM-pinned one: a CPU chain built from rjmcmc.py with the same random streams and the TDEM forward /
Jacobian of TdemBatch walks the same chain; cached state = from-scratch evaluation; synthetic data are fitted."""
import math
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import rj_emul
from geobipy_amd import rjmcmc
from geobipy_amd import rjmcmc_gpu as rg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OFFSET = (-13.0, 0.0, 2.0)
OPTIONS = dict(solve_gradient=True, maximum_number_of_layers=20, minimum_depth=1.0, maximum_depth=300.0, minimum_thickness=1.0,
               initial_relative_error=0.05, minimum_relative_error=0.005, maximum_relative_error=0.5,
               relative_error_proposal_variance=1e-6, additive_error_proposal_variance=1e-5, probability_of_birth=1.0 / 6.0,
               probability_of_death=1.0 / 6.0, probability_of_perturb=1.0 / 6.0, probability_of_no_change=0.5, covariance_scaling=0.5)


def _survey(B, seed=0):
    from geobipy_amd.tdem import TdemBatch, TdemSystem
    s = TdemSystem(os.path.join(GOLDEN, "SkytemLM.stm"))
    rng = np.random.default_rng(seed)
    K = 20
    sig, thk = np.ones((B, K)), np.zeros((B, K))
    sig[:, :3] = np.c_[10.0 ** rng.uniform(-2.5, -1.5, B), 10.0 ** rng.uniform(-1.2, -0.5, B), 10.0 ** rng.uniform(-2.5, -1.5, B)]
    thk[:, :2] = np.c_[rng.uniform(8, 25, B), rng.uniform(10, 40, B)]
    h = rng.uniform(30.0, 40.0, B)
    clean = TdemBatch(s, np.full(B, 3), sig, thk, h, OFFSET).forward().cpu().numpy()
    scale = np.sqrt(1e-3 / np.tile(s.off_time, s.n_components))
    add0 = 0.02 * np.abs(clean).min(axis=1).mean() / scale.min()          # additive level: a few % of the smallest gate
    std = np.sqrt((0.05 * clean) ** 2 + (add0 * scale) ** 2)
    data = clean + rng.normal(size=clean.shape) * std
    opts = dict(OPTIONS, initial_additive_error=add0, minimum_additive_error=add0 / 30.0, maximum_additive_error=add0 * 30.0)
    return s, h, data, scale, opts


def test_tdem_chains_equal_cpu_chains_with_the_same_seeds():
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    B, n_it = 3, 250
    s, h, data, scale, opts = _survey(B, seed=3)
    assert np.all(data > 0)
    dc = TdemDeviceChains(s, h, data, OFFSET, seed=77, **opts)
    assert np.allclose(dc.add_scale.cpu().numpy(), scale)

    class Engine:                       # TDEM forward / Jacobian of one sounding through TdemBatch (B = 1)
        def __init__(self, z):
            self.z = z

        def _batch(self, e, v):
            K = 20
            sg, th = np.ones((1, K)), np.zeros((1, K))
            sg[0, : v.size], th[0, : v.size - 1] = v, np.diff(np.r_[0.0, e])
            return TdemBatch(s, np.array([v.size]), sg, th, np.array([self.z]), OFFSET)

        def forward(self, e, v):
            return self._batch(e, v).forward().cpu().numpy()[0].copy()

        def sensitivity(self, e, v):
            return self._batch(e, v).sensitivity().cpu().numpy()[0][:, : v.size].copy()

    o = dc._o
    eo = dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge, p=[o.p_birth, o.p_death, o.p_perturb, o.p_none],
              rel_sd=o.rel_sd[0], rel_min=o.rel_min[0], rel_max=o.rel_max[0], add_sd=o.add_sd[0], add_min=o.add_min[0], add_max=o.add_max[0],
              alpha=o.alpha)
    sig0 = dc.sigma[:, 0].cpu().numpy()
    chains = []
    for b in range(B):
        sp = rjmcmc.StructurePrior(dc.K, opts["minimum_depth"], opts["maximum_depth"], opts["minimum_thickness"], eo["p"])
        vp = rjmcmc.ValuePrior(sig0[b], 10.0, 1.5, True)
        chains.append(rj_emul.Chain(eo, 77, b, Engine(h[b]), sp, vp, data[b], sig0[b], opts["initial_relative_error"],
                                    opts["initial_additive_error"], dc.n_depth_bins, dc.depth_bin_width, add_scale=scale))
        assert np.isclose(chains[b].misfit, float(dc.misfit[b]), rtol=1e-8) and np.isclose(chains[b].prior, float(dc.prior[b]), rtol=1e-12)
    acts, accs, ks = [], [], []
    prev = dc.n_accepted.cpu().numpy().copy()
    for it in range(n_it):
        dc.step()
        now = dc.n_accepted.cpu().numpy()
        acts.append(dc.action.cpu().numpy().copy()); accs.append(now - prev); ks.append(dc.k.cpu().numpy().copy())
        prev = now.copy()
        for c in chains:
            c.step(it)
    acts, accs, ks = np.array(acts), np.array(accs), np.array(ks)
    for b, c in enumerate(chains):
        tr = np.array(c.trace)
        assert np.array_equal(tr[:, 0], acts[:, b]) and np.array_equal(tr[:, 1], accs[:, b]) and np.array_equal(tr[:, 2], ks[:, b]), b
        assert np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-5)
    assert accs.sum() > 0.15 * accs.size and set(np.unique(acts)) == {0, 1, 2, 3}


def test_tdem_chains_fit_synthetic_soundings_and_stay_coherent():
    from geobipy_amd.tdem import TdemBatch, TdemDeviceChains
    B = 256
    s, h, data, scale, opts = _survey(B, seed=5)
    dc = TdemDeviceChains(s, h, data, OFFSET, seed=1, **opts)
    m0 = dc.misfit.clone()
    dc.run(1500)
    k = dc.k.cpu().numpy()
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64))
    tb = TdemBatch(s, k, dc.sigma.cpu().numpy(), thk.cpu().numpy(), h, OFFSET, data=data, relative_error=dc.rel.cpu().numpy(),
                   additive_error=dc.add.cpu().numpy())
    chi2, logl = tb.forward_loglike()
    assert torch.allclose(tb.predicted, dc.pred, rtol=1e-8, atol=0) and torch.allclose(chi2, dc.misfit, rtol=1e-7)
    assert torch.allclose(logl, dc.like, rtol=1e-9)
    n_ch = data.shape[1]
    print("TDEM misfit: start median", float(m0.median()), "-> after 1500 iterations", float(dc.misfit.median()), "channels", n_ch,
          "mean k", k.mean())
    assert float(dc.misfit.median()) < 2.0 * n_ch < float(m0.median())
    assert int(dc.k_hist.sum()) == 1500 * B and (k >= 2).mean() > 0.8
