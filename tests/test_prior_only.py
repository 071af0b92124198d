"""``ignore_likelihood``: the samplers without data (inversion/Inference1D.py:394, 519, 551, 596).  tests/prior_only.py says what
such a chain samples and which parts have a closed form.  This is the one check of the sampler that does not pass through the
repository's own stage emulation: the device chains against (a) analytic stationary laws and (b) an ensemble of the HOST sampler
(tests/golden/prior_only_host.npz, made by tests/golden/make_prior_only_host.py), which shares no random numbers with the device."""
import math
import os
import types

import numpy as np
import pytest

import prior_only as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ks_pvalue(x, cdf):
    from scipy.stats import kstest
    return kstest(x, cdf).pvalue


def _level_cdf(lo, hi, variance):
    """CDF of ln(level) under the analytic stationary density (prior_only.level_density), by quadrature."""
    from scipy.stats import norm
    s = math.sqrt(variance)
    x = np.linspace(math.log(lo), math.log(hi), 4001)
    z = norm.cdf((math.log(hi) - x) / s) - norm.cdf((math.log(lo) - x) / s)
    c = np.r_[0.0, np.cumsum(0.5 * (z[1:] + z[:-1]))]
    return lambda q: np.interp(q, x, c / c[-1])


# ---- CPU tier: the host sampler against the closed forms, the fixture, the object API -------------------------------------------
def test_host_fixed_structure_is_the_analytic_law():
    from scipy.stats import norm
    from geobipy_amd.inference import Inference1D
    vals, rels, adds, accepted, n = [], [], [], 0, 0
    for seed in range(80):                                   # 80 independent chains x 200 iterations (the level walk mixes in ~150)
        inf = Inference1D(prng=np.random.Generator(np.random.PCG64DXSM(500 + seed)), ignore_likelihood=True,
                          engine=po.NullEngine(po.N_CHANNELS), **po.FIXED)
        inf.initialize(types.SimpleNamespace(data=np.full(po.N_CHANNELS, 100.0), z=np.array([30.0])))
        mu = math.log(inf.halfspace[0])
        for i in range(200):
            inf.accept_reject()
            accepted += bool(inf.accepted)
            n += 1
            if i >= 150 and i % 10 == 0:
                vals.append(math.log(inf.state.values[0]) - mu)       # (i.i.d. draws: every proposal is accepted)
        assert inf.state.k == 1 and inf.state.like == 0.0 and inf.state.misfit == 0.0
        rels.append(math.log(inf.state.rel))
        adds.append(math.log(inf.state.add))
    assert accepted == n                                      # constant prior, no data: nothing is ever rejected
    assert _ks_pvalue(np.array(vals), norm(0.0, po.VALUE_SD).cdf) > 1e-3
    o = po.FIXED
    assert _ks_pvalue(np.array(rels), _level_cdf(o["minimum_relative_error"], o["maximum_relative_error"], o["relative_error_proposal_variance"])) > 1e-3
    assert _ks_pvalue(np.array(adds), _level_cdf(o["minimum_additive_error"], o["maximum_additive_error"], o["additive_error_proposal_variance"])) > 1e-3


def test_host_fixture_is_a_distribution_and_a_fresh_chain_agrees():
    f = np.load(os.path.join(GOLDEN, "prior_only_host.npz"))
    for key in ("k", "depth", "value", "rel", "add"):
        assert abs(f[key + "_p"].sum() - 1.0) < 1e-9 and f[key + "_se"].max() < 0.01
    # (NOT the uniform prior of the layer count: without the structure move's proposal terms in the ratio the reference's chain
    #  prefers few layers -- 62 % half-spaces, 36 % two layers, nothing beyond four with these options)
    assert f["k_p"][0] == 0.0 and f["k_p"][1:4].min() > 0.0 and f["k_p"][1] > f["k_p"][2] > f["k_p"][3]
    # the error levels of the full sampler follow the analytic law too (their walk does not see the model)
    o = po.OPTS
    for key, lo, hi, var in (("rel", o["minimum_relative_error"], o["maximum_relative_error"], o["relative_error_proposal_variance"]),
                             ("add", o["minimum_additive_error"], o["maximum_additive_error"], o["additive_error_proposal_variance"])):
        want = po.level_density(lo, hi, var)
        assert np.all(np.abs(f[key + "_p"] - want) <= 5.0 * f[key + "_se"] + 0.002), (key, f[key + "_p"], want)
    counts, acc = po.host_chain(77, 3000, 1000, 10)           # a short fresh chain: same harness, loose bars
    p = counts["k"] / counts["k"].sum()
    assert np.abs(p - f["k_p"]).max() < 0.25 and abs(acc - f["acceptance"].mean()) < 0.1


def test_host_infer_runs_the_prior_only_schedule():
    """Inference1D.infer with ignore_likelihood: burned in from the start with burned_in_iteration = n_markov_chains, 2 n_markov_chains + 1
    updates, never failed (inversion/Inference1D.py:388-389, 656)."""
    from geobipy_amd.inference import Inference1D
    n_mc = 40
    inf = Inference1D(prng=np.random.Generator(np.random.PCG64DXSM(9)), ignore_likelihood=True, engine=po.NullEngine(po.N_CHANNELS),
                      **dict(po.OPTS, n_markov_chains=n_mc))
    inf.initialize(types.SimpleNamespace(data=np.full(po.N_CHANNELS, 100.0), z=np.array([30.0])))
    assert inf.infer() is False
    assert inf.burned_in and inf.burned_in_iteration == n_mc and inf.iteration == 2 * n_mc + 1
    assert inf.posteriors.layers.sum() == 2 * n_mc + 1 if hasattr(inf.posteriors, "layers") else True
    assert np.array_equal(inf.observed, np.full(po.N_CHANNELS, 100.0)) and not inf.data.any()


def test_model_perturb_without_an_observation():
    """Model.perturb / proposal_probabilities with observation = None (model/Model.py:269, 352, 380): prior-only proposals."""
    from geobipy_amd import rjmcmc
    from geobipy_amd.model import Model, RectilinearMesh1D
    m = Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, 10.0, 40.0, np.inf]), values=np.array([0.01, 0.1, 0.03]))
    sp = rjmcmc.StructurePrior(8, 1.0, 150.0, 1.0, [1, 1, 1, 3])
    vp = rjmcmc.ValuePrior(0.02, 10.0, 1.5, True)
    m._structure_prior, m._value_prior, m._prng = sp, vp, np.random.Generator(np.random.PCG64DXSM(3))
    seen = set()
    for _ in range(200):
        rem, prop = m.perturb(None, alpha=1.0)
        fwd, rev = prop.proposal_probabilities(rem, None, alpha=1.0)
        seen.add(prop.mesh.action[0])
        assert np.all(np.isfinite(prop.values)) and prop.values.size == rem.values.size
        if prop.mesh.action[0] in ("insert", "delete"):
            assert np.isfinite(fwd) and np.isfinite(rev)
        else:
            assert (fwd, rev) == (1.0, 1.0)
    assert {"insert", "delete"} <= seen


# ---- GPU tier -----------------------------------------------------------------------------------------------------------------
def _device_chains(B, opts, seed):
    import torch
    from geobipy_amd import DeviceChains, FdemBatch, synthetic
    system = synthetic.syn10_system()
    nl, sigma, thk, height = synthetic.draw_models(B, 3, seed=11)
    data = FdemBatch(system, nl, sigma, thk, height).forward().cpu().numpy()
    assert (data > 0).all()
    o = {k: v for k, v in opts.items() if k != "n_markov_chains"}
    dc = DeviceChains(system, height, data, seed=seed, ignore_likelihood=True, **o)
    assert float(dc.t["data"].abs().max()) == 0.0 and torch.equal(dc.observed.cpu(), torch.as_tensor(data))
    return dc


def _snapshot(dc):
    t = dc.t
    return (t["k"].cpu().numpy().astype(np.int64), t["edges"].cpu().numpy(),
            (t["sigma"].log() - t["log_mean_prior"][:, None]).cpu().numpy(), t["rel"][:, 0].cpu().numpy(), t["add"][:, 0].cpu().numpy())


@pytest.mark.gpu
def test_device_fixed_structure_is_the_analytic_law():
    import torch
    from scipy.stats import chisquare, norm
    B, n_it = 8192, 400
    dc = _device_chains(B, po.FIXED, seed=5)
    dc.run(n_it)
    torch.cuda.synchronize()
    k, _, v, rel, add = _snapshot(dc)
    assert (k == 1).all()
    assert int(dc.n_accepted.min()) == n_it                    # constant prior, no data: nothing is ever rejected
    assert float(dc.like.abs().max()) == 0.0 and float(dc.misfit.abs().max()) == 0.0
    assert _ks_pvalue(v[:, 0], norm(0.0, po.VALUE_SD).cdf) > 1e-3
    o = po.FIXED
    for x, lo, hi, var in ((rel, o["minimum_relative_error"], o["maximum_relative_error"], o["relative_error_proposal_variance"]),
                           (add, o["minimum_additive_error"], o["maximum_additive_error"], o["additive_error_proposal_variance"])):
        counts = np.histogram(np.log(x), po.level_edges(lo, hi))[0]
        assert counts.sum() == B
        assert chisquare(counts, po.level_density(lo, hi, var) * B).pvalue > 1e-4, counts


@pytest.mark.gpu
def test_device_ensemble_equals_the_host_samplers():
    """8 192 device chains x 20 000 iterations against the host-sampler ensemble, cell by cell: layer count, interface depths,
    conductivities relative to the prior mean, error levels; and the error levels against their analytic law."""
    import torch
    f = np.load(os.path.join(GOLDEN, "prior_only_host.npz"))
    B = 8192
    dc = _device_chains(B, po.OPTS, seed=9)
    dc.run(5000)
    sums, n_snap = None, 0
    for _ in range(15):                                        # 15 snapshots, 1 000 iterations apart: 20 000 iterations in all
        dc.run(1000)
        torch.cuda.synchronize()
        c = po.reduce_states(*_snapshot(dc))
        sums = c if sums is None else {key: sums[key] + c[key] for key in c}
        n_snap += 1
    worst = {}
    for key in ("k", "depth", "value", "rel", "add"):
        p = sums[key] / sums[key].sum()
        # the device's standard error: B independent chains (the snapshots of a chain are correlated -- counted as one), cell by cell
        se = np.sqrt(f[key + "_se"] ** 2 + np.maximum(p * (1.0 - p), 1e-6) / B * (sums[key].sum() / (n_snap * B)))
        z = np.abs(p - f[key + "_p"]) / (se + 5e-4)
        worst[key] = float(z.max())
        assert z.max() < 5.0, (key, np.round(p, 4), np.round(f[key + "_p"], 4), np.round(z, 2))
    acc = float(dc.n_accepted.double().mean()) / dc.iteration
    assert abs(acc - f["acceptance"].mean()) < 5.0 * f["acceptance"].std(ddof=1) / math.sqrt(f["acceptance"].size) + 0.003, (acc, f["acceptance"].mean())
    o = po.OPTS
    for key, lo, hi, var in (("rel", o["minimum_relative_error"], o["maximum_relative_error"], o["relative_error_proposal_variance"]),
                             ("add", o["minimum_additive_error"], o["maximum_additive_error"], o["additive_error_proposal_variance"])):
        p = sums[key] / sums[key].sum()
        want = po.level_density(lo, hi, var)
        assert np.all(np.abs(p - want) <= 5.0 * np.sqrt(want / B) + 0.001), (key, p, want)
    print("prior-only: worst z by statistic", worst, "acceptance", acc)


@pytest.mark.gpu
def test_survey_driver_accepts_ignore_likelihood(tmp_path):
    from geobipy_amd import survey
    ds = survey.FdemData.read_csv(os.path.join(GOLDEN, "resolve_glacial_clean.csv"), os.path.join(GOLDEN, "resolve.stm"))
    ds = ds.subset(np.arange(24))
    n_mc = 300
    res = survey.infer(os.path.join(GOLDEN, "resolve_options_small"), data=ds, n_markov_chains=n_mc, ignore_likelihood=True,
                       results_directory=str(tmp_path), output=str(tmp_path / "summary.npz"))
    # Inference1D.py:388-389, 656: burned in from the start with burned_in_iteration = n_markov_chains, 2 n_markov_chains + 1 updates
    assert (res["status"] == 1).all() and (res["burned_in_iteration"] == n_mc).all() and (res["iterations"] == 2 * n_mc + 1).all()
    assert (res["misfit"] == 0.0).all()
    assert (res["layer_count_posterior"].sum(axis=1) == 2 * n_mc + 1).all()      # posteriors from the first update
    assert len(list(tmp_path.glob("*.h5"))) + len(list(tmp_path.glob("*.results.npz"))) >= 1
