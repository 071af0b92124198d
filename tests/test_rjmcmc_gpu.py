"""Device-resident rjMCMC (geobipy_amd/rjmcmc_gpu.py, csrc/gbp_rjmcmc.h).

CPU tier: the counter-based generator against its published known answers, the batched prior against rjmcmc.py.
GPU tier: every stage kernel against the host emulation (tests/rj_emul.py = rjmcmc.py + the same random streams);
state coherence after many steps; the posterior accumulators against a host replay; the ensembles against the
host sampler that reproduces the reference's chains."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

import rj_emul
from geobipy_amd import rjmcmc
from geobipy_amd import rjmcmc_gpu as rg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 12


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert philox_hex(0, (0, 0, 0, 0)) == ("6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8")
    assert philox_hex(0xFFFFFFFFFFFFFFFF, (0xFFFFFFFF,) * 4) == ("408f276d", "41c83b0e", "a20bc7c6", "6d5451fd")
    assert philox_hex(0x299F31D0_A4093822, (0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344)) == ("d16cfe09", "94fdcceb", "5001e420", "24126ea1")
    r = rj_emul.Rng(7, 3, 11, 0)
    u = np.array([r.uniform() for _ in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.003
    z = np.array([v for j in range(10000) for v in rj_emul.normal_pair(7, 3, 11, 1, j)])
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1.0) < 0.03 and abs(np.mean(z ** 4) - 3.0) < 0.15


def philox_hex(seed, ctr):
    return tuple("%08x" % v for v in rj_emul.philox(seed, *ctr))


def _random_models(rng, B, kmax=K - 1, pad=K):
    ks = rng.integers(1, kmax + 1, size=B)
    edges = np.full((B, pad), np.inf)
    sigma = np.ones((B, pad))
    models = []
    for b, k in enumerate(ks):
        e = np.cumsum(rng.uniform(1.5, 14.0, size=k - 1)) + 0.3
        v = 10.0 ** rng.uniform(-3, 0, size=k)
        edges[b, : k - 1], sigma[b, :k] = e, v
        models.append((e, v))
    return ks, edges, sigma, models


def test_batched_priors_match_host():
    rng = np.random.default_rng(5)
    B = 64
    ks, edges, sigma, models = _random_models(rng, B)
    vp = rjmcmc.ValuePrior(0.01, 10.0, 1.5, True)
    sp = rjmcmc.StructurePrior(K, 1.0, 300.0, 1.0, [1, 1, 1, 3])
    t = torch.as_tensor
    lp = rg.model_log_prior(t(edges), t(sigma), t(ks.astype(np.int64)), K, vp.gradient_precision)
    for b, (e, v) in enumerate(models):
        np.testing.assert_allclose(float(lp[b]), rjmcmc.model_log_prior(sp, vp, e, v), rtol=1e-12)
    x = torch.tensor([0.001, 0.01, 0.5, 0.6], dtype=torch.float64)
    p = rjmcmc.ErrorPrior(0.005, 0.5, 1e-4)
    for g, xi in zip(rg.log_uniform_prior(x, 0.005, 0.5).numpy(), x.numpy()):
        assert g == p.log_prior(xi) or np.isclose(g, p.log_prior(xi), rtol=1e-14)


# ------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------
def _options():
    from test_rjmcmc import RESOLVE_OPTIONS
    return {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}


def _chains(B, seed, exact=False, n_it=0, **kw):
    from geobipy_amd import FdemSystem
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    s = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    o = _options()
    o.update(kw.pop("options", {}))
    dc = rg.DeviceChains(s, np.full(B, float(d["z"])), np.tile(d["data"], (B, 1)), seed=seed, exact_jacobian=exact, **o, **kw)
    return d, s, dc.run(n_it)


def _emul_options(dc):
    o = dc._o
    return dict(K=dc.K, min_width=o.min_width, min_edge=o.min_edge, max_edge=o.max_edge,
                p=[o.p_birth, o.p_death, o.p_perturb, o.p_none], rel_sd=o.rel_sd[0], rel_min=o.rel_min[0], rel_max=o.rel_max[0],
                add_sd=o.add_sd[0], add_min=o.add_min[0], add_max=o.add_max[0], alpha=o.alpha,
                height_scale=o.height_scale, height_half_width=o.height_half_width)


def _host_priors(dc, b):
    o = dc.o
    sp = rjmcmc.StructurePrior(dc.K, o["minimum_depth"], o["maximum_depth"], o["minimum_thickness"],
                               [o["probability_of_birth"], o["probability_of_death"], o["probability_of_perturb"],
                                o["probability_of_no_change"]])
    vp = rjmcmc.ValuePrior(math.exp(float(dc.log_mean_prior[b])), o["factor"], o["gradient_standard_deviation"], o["solve_gradient"],
                           bool(o.get("solve_parameter", False)), o.get("parameter_limits"))
    return sp, vp


def _load_random_state(dc, rng, kmax):
    """Overwrite the chain state with random valid models (and wide error levels so that the error moves get rejected
    sometimes)."""
    ks, edges, sigma, models = _random_models(rng, dc.B, kmax=kmax, pad=dc.K)
    dc.k.copy_(torch.as_tensor(ks.astype(np.int32)))
    dc.edges.copy_(torch.as_tensor(edges))
    dc.sigma.copy_(torch.as_tensor(sigma))
    dc.rel[:, 0] = torch.as_tensor(rng.uniform(0.0012, 0.4, dc.B))
    dc.add[:, 0] = torch.as_tensor(rng.uniform(3.05, 19.5, dc.B))
    return ks, models


@pytest.mark.gpu
def test_rj_entries_refuse_bad_arguments():
    """gbp_rj_* validate what they can on the host: sizes, NULL pointers, the option / buffer combinations."""
    import copy
    from geobipy_amd import _lib
    lib = _lib.load()
    _, _, dc = _chains(8, 1)

    def status(o=None, c=None, run=True):
        o, c = o or dc._o, c or dc._c
        return lib.gbp_rj_run(dc._h.ptr, o, c, 0, 1, 0, None) if run else lib.gbp_rj_propose(o, c, 0, None)

    def opt(**kw):
        o = _lib.RjOptions.from_buffer_copy(dc._o)
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def chains(**kw):
        c = _lib.RjChains.from_buffer_copy(dc._c)
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    assert status() == 0
    for bad in (opt(max_layers=1), opt(max_layers=65), opt(n_channels=0), opt(min_width=0.0), opt(max_edge=0.5),
                opt(n_channels=10), opt(n_depth_bins=0)):
        assert status(o=bad) == -1, "GBP_ERR_INVALID_ARG expected"
        assert b"" != lib.gbp_last_error()
    for bad in (chains(sigma=None), chains(J_p=None), chains(nl_b=None), chains(B=-1), chains(hitmap=dc.k_hist.data_ptr())):
        assert status(c=bad) == -1
    assert lib.gbp_rj_run(None, dc._o, dc._c, 0, 1, 0, None) == -1
    assert status(o=opt(schedule=1, n_markov_chains=10), c=chains(status=None)) == -1
    assert status(c=chains(B=0)) == 0                                   # an empty block is fine
    assert lib.gbp_rj_flush_posteriors(dc._o, dc._c, None) == 0         # no hit map: nothing to do


@pytest.mark.gpu
def test_device_random_streams_match_the_emulation():
    from geobipy_amd import _lib
    n = 64
    u, z = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
    for seed, chain, it, st in ((0, 0, 0, 0), (12345678901234567, 4100, 77, 1), (2 ** 64 - 1, 65535, 2 ** 31, 2)):
        _lib.check(_lib.load().gbp_rj_debug_random(seed, chain, it, st, n, u.data_ptr(), z.data_ptr(), None))
        r = rj_emul.Rng(seed, chain, it & 0xFFFFFFFF, st)
        assert np.array_equal(u.cpu().numpy(), [r.uniform() for _ in range(n)])
        ref = [v for j in range(n // 2) for v in rj_emul.normal_pair(seed, chain, it & 0xFFFFFFFF, st, j)]
        assert np.allclose(z.cpu().numpy(), ref, rtol=1e-14, atol=1e-15)


@pytest.mark.gpu
def test_propose_kernel_matches_the_emulation():
    """Structural move, remapping, layer-bucket masks and error proposals of 2048 chains, two iterations, one of them
    with the error proposal widths blown up so that the redraw / give-up path is exercised."""
    from geobipy_amd import _lib
    _, _, dc = _chains(2048, 99, options=dict(relative_error_proposal_variance=400.0, additive_error_proposal_variance=1.0,
                                              maximum_depth=150.0))
    rng = np.random.default_rng(8)
    ks, models = _load_random_state(dc, rng, kmax=dc.K)
    eo = _emul_options(dc)
    seen = set()
    kept = 0
    for it in (5, 6):
        _lib.check(_lib.load().gbp_rj_propose(dc._o, dc._c, it, None))
        act, k_r = dc.action.cpu().numpy(), dc.k_r.cpu().numpy()
        e_r, s_r, t_r = dc.edges_r.cpu().numpy(), dc.sigma_r.cpu().numpy(), dc.thk_r.cpu().numpy()
        rel_p, add_p = dc.rel_p[:, 0].cpu().numpy(), dc.add_p[:, 0].cpu().numpy()
        nl_a, nl_c = dc.nl_a.cpu().numpy(), dc.nl_c.cpu().numpy()
        rel, add = dc.rel[:, 0].cpu().numpy(), dc.add[:, 0].cpu().numpy()
        for b, (e, v) in enumerate(models):
            a, idx, val, ee, ss, rp, ap = rj_emul.propose(eo, 99, b, it, e, v, rel[b], add[b])
            assert act[b] == a and k_r[b] == ss.size, (b, act[b], a)
            assert np.allclose(e_r[b, : ss.size - 1], ee, rtol=1e-14) and np.all(np.isinf(e_r[b, ss.size - 1:]))
            assert np.array_equal(s_r[b, : ss.size], ss) and np.all(s_r[b, ss.size:] == 1.0)
            assert np.allclose(t_r[b, : ss.size - 1], np.diff(np.r_[0.0, ee]), rtol=1e-13) and not t_r[b, ss.size - 1:].any()
            assert np.isclose(rel_p[b], rp, rtol=1e-13) and np.isclose(add_p[b], ap, rtol=1e-13)
            kept += rp == rel[b]
            bucket = 1 + (0 if ss.size <= 8 else 1)
            want_a = np.zeros(3, dtype=int)
            want_c = np.zeros(3, dtype=int)
            if a != rjmcmc.NONE:
                want_a[0] = want_a[bucket] = ss.size
            if a in (rjmcmc.INSERT, rjmcmc.DELETE):
                want_c[0] = want_c[bucket] = ss.size
            assert np.array_equal(nl_a[:, b], want_a) and np.array_equal(nl_c[:, b], want_c)
            seen.add(a)
    assert seen == {0, 1, 2, 3} and kept > 20          # all four moves, and the give-up path of the error proposal
    # the three implementations of the draws write the same bits: one wave per chain, one thread per chain, and the default
    # (one thread per chain, rows staged through LDS)
    names = ("action", "k_r", "nl_a", "nl_c", "edges_r", "sigma_r", "thk_r", "rel_p", "add_p")
    out = {}
    for variant, code in (("wave", 2), ("thread", 1), ("default", 0)):
        for n in names:
            getattr(dc, n).fill_(-7)
        _lib.check(_lib.load().gbp_rj_debug_propose_variant(dc._o, dc._c, 6, code, None))
        out[variant] = {n: getattr(dc, n).clone() for n in names}
    for n in names:
        assert torch.equal(out["wave"][n], out["thread"][n]) and torch.equal(out["default"][n], out["thread"][n]), n
    assert np.array_equal(out["wave"]["action"].cpu().numpy(), act)


@pytest.mark.gpu
def test_newton_kernel_matches_the_emulation():
    from geobipy_amd import _lib
    _, _, dc = _chains(512, 3)
    rng = np.random.default_rng(21)
    ks, models = _load_random_state(dc, rng, kmax=dc.K)
    B, N, Kp = dc.B, dc.N, dc.K
    # a remapped model = the state itself; half the chains read the "changed" buffers, half the carried ones
    dc.edges_r.copy_(dc.edges); dc.sigma_r.copy_(dc.sigma); dc.k_r.copy_(dc.k)
    action = rng.integers(0, 4, B).astype(np.int32)
    dc.action.copy_(torch.as_tensor(action))
    J_a, J_b = rng.normal(size=(B, N, Kp)) * 40.0, rng.normal(size=(B, N, Kp)) * 40.0
    data = dc.data.cpu().numpy().copy()
    data[::3, 4] = -1.0                                                      # inactive channels
    dc.data.copy_(torch.as_tensor(data))
    p_a, p_b = data * (1 + 0.1 * rng.normal(size=data.shape)), data * (1 + 0.1 * rng.normal(size=data.shape))
    dc.J.copy_(torch.as_tensor(J_a)); dc.J_r.copy_(torch.as_tensor(J_b))
    dc.pred.copy_(torch.as_tensor(p_a)); dc.pred_r.copy_(torch.as_tensor(p_b))
    _lib.check(_lib.load().gbp_rj_newton(dc._o, dc._c, 17, None))
    lp, sp_, ch = dc.log_prop.cpu().numpy(), dc.sigma_p.cpu().numpy(), dc.chol.cpu().numpy()
    rel, add = dc.rel[:, 0].cpu().numpy(), dc.add[:, 0].cpu().numpy()
    eo = _emul_options(dc)
    for b, (e, v) in enumerate(models):
        _, vp = _host_priors(dc, b)
        k = v.size
        J, pred = (J_b, p_b) if action[b] else (J_a, p_a)
        ref_lp, ref_C = rj_emul.newton(eo, 3, b, 17, vp, e, v, J[b], pred[b], data[b], rel[b], add[b])
        assert np.allclose(np.tril(ch[b, :k, :k]), ref_C, rtol=1e-9, atol=1e-12), b
        assert np.allclose(lp[b, :k], ref_lp, rtol=1e-8, atol=1e-9), (b, k, lp[b, :k] - ref_lp)
        assert np.allclose(sp_[b, :k], np.exp(ref_lp), rtol=1e-8) and np.all(sp_[b, k:] == 1.0) and not lp[b, k:].any()


@pytest.mark.gpu
def test_accept_kernel_matches_the_emulation():
    """Run propose + the remapped-model kernels + newton + the proposal kernels through the library on random states,
    then check the accept kernel's ratio, decision and state update against the emulation fed with the device's own
    intermediate buffers."""
    from geobipy_amd import _lib
    lib = _lib.load()
    _, _, dc = _chains(1024, 41, exact=True)
    rng = np.random.default_rng(2)
    ks, models = _load_random_state(dc, rng, kmax=10)
    dc.rel.fill_(0.05); dc.add.fill_(5.0)
    B, N, Kp, h = dc.B, dc.N, dc.K, dc._h.ptr
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64)).contiguous()
    # a coherent carried state: prediction / Jacobian / likelihood / prior of the random models
    _lib.check(lib.gbp_fdem_forward_loglike(h, B, Kp, dc.k.data_ptr(), dc.sigma.data_ptr(), thk.data_ptr(), dc.height.data_ptr(),
                                            dc.data.data_ptr(), dc.rel.data_ptr(), dc.add.data_ptr(), dc.pred.data_ptr(),
                                            dc.misfit.data_ptr(), dc.like.data_ptr(), None))
    _lib.check(lib.gbp_fdem_sensitivity_ex(h, B, Kp, dc.k.data_ptr(), dc.sigma.data_ptr(), thk.data_ptr(), dc.height.data_ptr(),
                                           dc.J.data_ptr(), Kp, 1, None))
    o = dc.o
    dc.prior.copy_(rg.model_log_prior(dc.edges, dc.sigma, dc.k.to(torch.int64), Kp, dc.gradient_precision)
                   + rg.log_uniform_prior(dc.rel[:, 0], o["minimum_relative_error"], o["maximum_relative_error"])
                   + rg.log_uniform_prior(dc.add[:, 0], o["minimum_additive_error"], o["maximum_additive_error"]))
    before = {n: getattr(dc, n).clone() for n in ("k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "misfit")}
    it = 9
    dc.iteration = it
    _lib.check(lib.gbp_rj_propose(dc._o, dc._c, it, None))
    dc2_state = {n: getattr(dc, n).cpu().numpy().copy() for n in ("action", "k_r", "edges_r", "sigma_r", "rel_p", "add_p")}
    dc.run(1)                                                   # the same iteration, complete
    g = lambda n: getattr(dc, n).cpu().numpy()
    for n, v in dc2_state.items():
        assert np.array_equal(g(n), v, equal_nan=True), n       # run() proposed exactly what the stage call did
    action, k_r, e_r, s_r = g("action"), g("k_r"), g("edges_r"), g("sigma_r")
    log_prop, chol, J_p, pred_p = g("log_prop"), g("chol"), g("J_p"), g("pred_p")
    rel_p, add_p, like_p, misfit_p, log_ratio = g("rel_p")[:, 0], g("add_p")[:, 0], g("like_p"), g("misfit_p"), g("log_ratio")
    data, n_accepted = g("data"), g("n_accepted")
    eo = _emul_options(dc)
    n_acc = n_jump = 0
    bk = {n: v.cpu().numpy() for n, v in before.items()}
    for b in range(B):
        sp, vp = _host_priors(dc, b)
        k = k_r[b]
        C = np.tril(chol[b, :k, :k])
        ref_misfit, ref_like = rjmcmc.gauss_loglike(pred_p[b], data[b], np.sqrt((rel_p[b] * data[b]) ** 2 + add_p[b] ** 2))
        if action[b] in (0, 3):                # proposals that keep their dimension: chi^2 / logL from the fused forward kernel
            assert np.isclose(like_p[b], ref_like, rtol=1e-12) and np.isclose(misfit_p[b], ref_misfit, rtol=1e-11)
        ref_lr, ref_acc, ref_prior = rj_emul.accept(eo, 41, b, it, sp, vp, action[b], e_r[b, : k - 1], s_r[b, :k], log_prop[b, :k], C,
                                                    J_p[b], pred_p[b], data[b], rel_p[b], add_p[b], ref_like, bk["prior"][b],
                                                    bk["like"][b])
        assert (np.isclose(log_ratio[b], ref_lr, rtol=1e-7, atol=1e-6, equal_nan=True)
                or (np.isneginf(ref_lr) and np.isneginf(log_ratio[b]))), (b, action[b], log_ratio[b], ref_lr)
        margin = abs(ref_lr - np.log(rj_emul.u53(*rj_emul.philox(41, b, it, 2, 0)[:2])))
        accepted = int(n_accepted[b]) == 1
        assert accepted == ref_acc or margin < 1e-5
        n_acc += accepted
        n_jump += action[b] in (1, 2)
        if accepted:
            assert int(dc.k[b]) == k
            assert np.array_equal(dc.edges[b].cpu().numpy(), e_r[b]) and np.array_equal(dc.sigma[b].cpu().numpy(), g("sigma_p")[b])
            assert np.array_equal(dc.pred[b].cpu().numpy(), pred_p[b]) and np.isclose(float(dc.like[b]), ref_like, rtol=1e-12)
            assert np.isclose(float(dc.misfit[b]), ref_misfit, rtol=1e-11) and np.isclose(float(dc.prior[b]), ref_prior, rtol=1e-12)
            assert float(dc.rel[b, 0]) == rel_p[b] and float(dc.add[b, 0]) == add_p[b]
            Jsrc = {0: bk["J"][b], 1: J_p[b], 2: J_p[b], 3: g("J_r")[b]}[int(action[b])]
            assert np.array_equal(dc.J[b].cpu().numpy(), Jsrc)
        else:
            for n in before:
                assert np.array_equal(getattr(dc, n)[b].cpu().numpy(), bk[n][b], equal_nan=True), (b, n)
    assert n_acc > 50 and n_jump > 100


@pytest.mark.gpu
def test_device_chains_initialise_like_the_reference():
    d, s, dc = _chains(4, 0)
    assert np.allclose(dc.sigma[:, 0].cpu().numpy(), d["halfspace"].item(), rtol=1e-14)
    assert np.allclose(dc.misfit.cpu().numpy(), d["cur_misfit"][0], rtol=1e-9)
    assert np.allclose(dc.prior.cpu().numpy(), d["cur_prior"][0], rtol=1e-12)
    assert np.allclose(dc.like.cpu().numpy(), d["cur_like"][0], rtol=1e-9)
    assert np.allclose(dc.J[0, :, :1].cpu().numpy(), d["cur_J"][0][:, :1], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_device_chains_state_is_coherent_after_many_steps():
    """After 300 iterations every chain's cached prediction / misfit / likelihood / prior equals a from-scratch
    evaluation of its current model, every model satisfies the structural constraints, and the run is reproducible."""
    from geobipy_amd import FdemBatch
    d, s, dc = _chains(256, 11, n_it=300)
    k = dc.k.cpu().numpy()
    e, sig = dc.edges.cpu().numpy(), dc.sigma.cpu().numpy()
    assert k.min() >= 1 and k.max() <= dc.K and len(np.unique(k)) > 2
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64)).cpu().numpy()
    for b in range(dc.B):
        w = thk[b, : k[b] - 1]
        assert np.all(w > dc.min_width) and np.all(np.isinf(e[b, k[b] - 1:])) and np.all(sig[b, k[b]:] == 1.0)
        if k[b] > 1:
            assert e[b, 0] > dc.min_edge and e[b, k[b] - 2] < dc.max_edge
    fb = FdemBatch(s, k, sig, thk, dc.height.cpu().numpy(), data=dc.data.cpu().numpy(),
                   relative_error=dc.rel[:, 0].cpu().numpy(), additive_error=dc.add[:, 0].cpu().numpy(),
                   waves=dc._o.forward_waves)            # the summation order and the abscissa windows the chains ran with
    chi2, logl = fb.forward_loglike()
    # proposals that keep their dimension get prediction / chi^2 / logL from the fused forward kernel (bit-equal to this
    # evaluation); the others from the Jacobian pass and the accept kernel (same values, different summation order)
    assert torch.allclose(fb.predicted, dc.pred, rtol=1e-11, atol=1e-9) and torch.allclose(chi2, dc.misfit, rtol=1e-9)
    assert torch.allclose(logl, dc.like, rtol=1e-10) and (fb.predicted == dc.pred).all(dim=1).float().mean() > 0.3
    o = dc.o
    prior = (rg.model_log_prior(dc.edges, dc.sigma, dc.k.to(torch.int64), dc.K, dc.gradient_precision)
             + rg.log_uniform_prior(dc.rel[:, 0], o["minimum_relative_error"], o["maximum_relative_error"])
             + rg.log_uniform_prior(dc.add[:, 0], o["minimum_additive_error"], o["maximum_additive_error"]))
    assert torch.allclose(prior, dc.prior, rtol=1e-12, atol=0)
    assert int(dc.k_hist.sum()) == 300 * dc.B
    assert torch.all(dc.best_posterior >= dc.like + dc.prior)
    _, _, dc2 = _chains(256, 11, n_it=300)
    for n in ("k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "n_accepted", "k_hist", "edge_hist"):
        assert torch.equal(getattr(dc, n), getattr(dc2, n)), n
    _, _, dc3 = _chains(256, 12, n_it=300)
    assert not torch.equal(dc.sigma, dc3.sigma)


@pytest.mark.gpu
def test_deep_chains_stay_coherent():
    """Chains started from random models with up to 30 layers (the one-wave-per-chain newton / accept kernels and the deep
    Jacobian bucket): after 60 iterations the carried prediction / misfit / likelihood / prior still describe the carried
    model, and layers have been both added and removed."""
    from geobipy_amd import FdemBatch, _lib
    lib = _lib.load()
    _, s, dc = _chains(512, 8, exact=True)
    rng = np.random.default_rng(12)
    ks0, _ = _load_random_state(dc, rng, kmax=dc.K)
    dc.rel.fill_(0.05); dc.add.fill_(5.0)
    B, Kp, h = dc.B, dc.K, dc._h.ptr
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64)).contiguous()
    _lib.check(lib.gbp_fdem_forward_loglike_ex(h, B, Kp, dc.k.data_ptr(), dc.sigma.data_ptr(), thk.data_ptr(), dc.height.data_ptr(),
                                               dc.data.data_ptr(), dc.rel.data_ptr(), dc.add.data_ptr(), dc.pred.data_ptr(),
                                               dc.misfit.data_ptr(), dc.like.data_ptr(), dc._o.forward_waves, None))
    _lib.check(lib.gbp_fdem_sensitivity_ex(h, B, Kp, dc.k.data_ptr(), dc.sigma.data_ptr(), thk.data_ptr(), dc.height.data_ptr(),
                                           dc.J.data_ptr(), Kp, 1, None))
    o = dc.o
    full_prior = lambda: (rg.model_log_prior(dc.edges, dc.sigma, dc.k.to(torch.int64), Kp, dc.gradient_precision)
                          + rg.log_uniform_prior(dc.rel[:, 0], o["minimum_relative_error"], o["maximum_relative_error"])
                          + rg.log_uniform_prior(dc.add[:, 0], o["minimum_additive_error"], o["maximum_additive_error"]))
    dc.prior.copy_(full_prior())
    dc.best_posterior.copy_(dc.prior + dc.like)
    dc.run(60)
    k = dc.k.cpu().numpy()
    n_acc = dc.n_accepted.cpu().numpy()
    print("deep chains:", (k > 8).sum(), "changed k:", (k != ks0).sum(), "accepted moves of deep chains:", n_acc[ks0 > 8].sum())
    assert (k > 8).sum() > 100 and (k != ks0).sum() > 30 and (k > ks0).any() and (k < ks0).any() and n_acc[ks0 > 8].sum() > 100
    thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64))
    assert torch.all(torch.where(thk > 0, thk, torch.full_like(thk, 9.0)) > dc.min_width)
    fb = FdemBatch(s, k, dc.sigma.cpu().numpy(), thk.cpu().numpy(), dc.height.cpu().numpy(), data=dc.data.cpu().numpy(),
                   relative_error=dc.rel[:, 0].cpu().numpy(), additive_error=dc.add[:, 0].cpu().numpy(),
                   waves=dc._o.forward_waves)
    chi2, logl = fb.forward_loglike()
    assert torch.allclose(fb.predicted, dc.pred, rtol=1e-10, atol=1e-8) and torch.allclose(chi2, dc.misfit, rtol=1e-8)
    assert torch.allclose(logl, dc.like, rtol=1e-9) and torch.allclose(full_prior(), dc.prior, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_device_chains_do_not_depend_on_the_sharding():
    """The random streams are keyed by the global chain index and forward_waves pins the summation order of the forward
    kernels: a block of 96 soundings run as one DeviceChains or as three shards of 32 gives bit-identical chains."""
    _, _, whole = _chains(96, 11, n_it=400, forward_waves=4)
    for first in (0, 32, 64):
        _, _, part = _chains(32, 11, n_it=400, first_chain=first, forward_waves=4)
        for n in ("k", "edges", "sigma", "rel", "add", "pred", "like", "prior", "n_accepted", "k_hist", "edge_hist", "best_sigma"):
            assert torch.equal(getattr(whole, n)[first:first + 32], getattr(part, n)), (first, n)


@pytest.mark.gpu
def test_large_blocks_fork_the_proposal_evaluations_without_changing_the_chains():
    """From ~32 k chains on, gbp_rj_run evaluates the dimension-changing proposals on a side stream next to the fused forward
    kernel (disjoint chains, disjoint rows).  A block of 33 000 chains -- forked -- against the same chains run as two
    shards below the threshold -- one stream: bit-identical states."""
    B, half = 33000, 16500
    _, _, whole = _chains(B, 21, n_it=40)
    for first in (0, half):
        _, _, part = _chains(half, 21, n_it=40, first_chain=first)
        for n in ("k", "edges", "sigma", "rel", "add", "pred", "J", "like", "prior", "n_accepted", "k_hist", "best_sigma"):
            assert torch.equal(getattr(whole, n)[first:first + half], getattr(part, n)), (first, n)
    assert int(whole.n_accepted.sum()) > B


@pytest.mark.gpu
def test_posterior_accumulators_match_a_host_replay():
    """The device accumulators against inference.Posteriors -- the host rule that reproduces the reference's own posterior
    counts bit for bit (tests/test_rjmcmc.py) -- fed with the chains' states iteration by iteration: layer-count
    histogram, interface-depth histogram, conductivity-depth hit map (also the default 250-bin value axis)."""
    from geobipy_amd.inference import Posteriors
    for nv in (50, 250):
        _, _, dc = _chains(16, 4, exact=True, hitmap=True, n_value_bins=nv)
        dc.run(200, accumulate=False)                            # burn-in: nothing is accumulated
        assert int(dc.k_hist.sum()) == 0 and int(dc.edge_hist.sum()) == 0 and int(dc.hitmap.sum()) == 0
        o = dc.o
        mean = np.exp(dc.log_mean_prior.cpu().numpy())
        posts = [Posteriors(dc.K, o["maximum_depth"], o["minimum_thickness"], mean[b], o["factor"], n_value_bins=nv,
                            relative_error_bounds=(o["minimum_relative_error"], o["maximum_relative_error"]),
                            additive_error_bounds=(o["minimum_additive_error"], o["maximum_additive_error"])) for b in range(dc.B)]
        assert posts[0].depth_centres.size == dc.n_depth_bins and np.isclose(posts[0].value_edges[-1], dc.value_half_width, rtol=1e-14)
        for _ in range(150):
            dc.step()
            k, e, s = dc.k.cpu().numpy(), dc.edges.cpu().numpy(), dc.sigma.cpu().numpy()
            rel, add = dc.rel[:, 0].cpu().numpy(), dc.add[:, 0].cpu().numpy()
            for b in range(dc.B):
                posts[b].update(e[b, : k[b] - 1], s[b, : k[b]], rel[b], add[b])
        assert np.array_equal(dc.k_hist.cpu().numpy(), np.stack([p.n_cells for p in posts]))
        assert np.array_equal(dc.edge_hist.cpu().numpy(), np.stack([p.edges for p in posts])) and dc.edge_hist.sum() > 0
        assert np.array_equal(dc.hitmap.cpu().numpy(), np.stack([p.values for p in posts]))    # [B, value, depth] both
        assert np.array_equal(dc.rel_hist[:, 0].cpu().numpy(), np.stack([p.relative_error for p in posts]))
        assert np.array_equal(dc.add_hist[:, 0].cpu().numpy(), np.stack([p.additive_error for p in posts]))


@pytest.mark.gpu
def test_abscissa_window_does_not_change_the_chains():
    """The default per-chain abscissa window (1e-10 ppm): predictions and Jacobians move by < 1e-10 ppm against the full
    120-point sums (hankel_eps_ppm=0), so 300 iterations take the same decisions and end in the same models to 1e-8."""
    _, _, a = _chains(64, 31, exact=True, n_it=300, hankel_eps_ppm=0.0)
    _, _, w = _chains(64, 31, exact=True, n_it=300)
    assert w._h.bins is not None and a._h.bins is None
    assert torch.equal(a.k, w.k) and torch.equal(a.n_accepted, w.n_accepted) and torch.equal(a.k_hist, w.k_hist)
    assert torch.allclose(a.sigma, w.sigma, rtol=1e-8) and torch.allclose(a.misfit, w.misfit, rtol=1e-8)


@pytest.mark.gpu
def test_reference_schedule_on_the_device_matches_a_host_replay():
    """Per-sounding burn-in / stop rule of Inference1D.update / infer evaluated in the accept kernel: burn-in iteration,
    status, the posterior that restarts at burn-in, the frozen final state -- against a replay of the rule on the host
    from the per-iteration misfits.  A few soundings carry data no layered earth fits, so that they fail to burn in."""
    n_mc, burn_min, B = 300, 100, 48
    d, s, dc = _chains(B, 77, exact=True, reference_schedule=True, burn_in_min_iterations=burn_min,
                       options=dict(n_markov_chains=n_mc))
    rng = np.random.default_rng(1)
    data = np.tile(d["data"], (B, 1))
    data[:6] *= rng.uniform(0.4, 2.5, size=(6, data.shape[1]))
    dc.data.copy_(torch.as_tensor(data))
    dc._initialize()
    bi = np.full(B, -1)
    status = np.zeros(B, dtype=int)
    k_hist = np.zeros((B, dc.K + 1), dtype=int)
    final = {}
    n_active = (data > 0).sum(axis=1)
    it1 = 0
    while it1 < 2 * n_mc + 2:
        dc.step()
        it1 += 1
        mis, k = dc.misfit.cpu().numpy(), dc.k.cpu().numpy()
        sig = dc.sigma.cpu().numpy()
        for b in range(B):
            if status[b]:
                assert np.array_equal(sig[b], final[b]), (b, it1)         # frozen
                continue
            if bi[b] < 0 and it1 > burn_min and mis[b] < n_active[b]:
                bi[b] = it1
                k_hist[b] = 0
            k_hist[b, k[b]] += 1
            if bi[b] >= 0 and it1 > n_mc + bi[b]:
                status[b] = 1
            elif bi[b] < 0 and it1 >= n_mc:
                status[b] = 2
            if status[b]:
                final[b] = sig[b].copy()
    assert np.array_equal(dc.burned_in_iteration.cpu().numpy(), bi) and np.array_equal(dc.status.cpu().numpy(), status)
    assert np.array_equal(dc.k_hist.cpu().numpy(), k_hist)
    assert (status == 1).sum() > B // 2 and (status == 2).sum() >= 3 and not (status == 0).any()
    assert np.all(k_hist[status == 1].sum(axis=1) == n_mc + 2) and np.all(k_hist[status == 2].sum(axis=1) == n_mc)
    assert torch.all(dc.best_posterior[dc.status == 1] <= (dc.like + dc.prior)[dc.status == 1] + 1e300)
    # infer(): the same run driven by the status flags
    _, _, dc2 = _chains(B, 77, exact=True, reference_schedule=True, burn_in_min_iterations=burn_min, options=dict(n_markov_chains=n_mc))
    dc2.data.copy_(torch.as_tensor(data))
    dc2._initialize()
    assert dc2.infer(check_every=64, compact_below=0.0) == (status == 2).sum()
    for n in ("k", "sigma", "k_hist", "edge_hist", "burned_in_iteration", "status", "best_sigma"):
        assert torch.equal(getattr(dc, n), getattr(dc2, n)), n
    # ... and with the block re-packed whenever chains have finished: the same chains, row for row
    _, _, dc3 = _chains(B, 77, exact=True, reference_schedule=True, burn_in_min_iterations=burn_min, hitmap=True, n_value_bins=20,
                        options=dict(n_markov_chains=n_mc))
    dc3.data.copy_(torch.as_tensor(data))
    dc3._initialize()
    sizes = []
    run0 = dc3.run
    dc3.run = lambda n, accumulate=True: (sizes.append(dc3._c.B), run0(n, accumulate))[1]
    assert dc3.infer(check_every=16, compact_below=0.95, min_rows=2) == (status == 2).sum()
    assert min(sizes) < B // 2 and dc3._c.B == B and dc3.k.shape[0] == B
    for n in ("k", "edges", "sigma", "rel", "add", "pred", "misfit", "k_hist", "edge_hist", "rel_hist", "burned_in_iteration", "status",
              "best_sigma", "n_accepted", "data"):
        assert torch.equal(getattr(dc, n), getattr(dc3, n)), n
    assert int(dc3.hitmap.sum()) == int(dc3.k_hist.sum()) * dc3.n_depth_bins


@pytest.mark.gpu
@pytest.mark.parametrize("priors", [dict(), dict(solve_parameter=True, solve_gradient=False),
                                    dict(maximum_number_of_layers=4, probability_of_birth=0.4),
                                    dict(parameter_limits=[0.02, 0.3])])
def test_device_chains_equal_cpu_chains_with_the_same_seeds(priors):
    """BASELINE config 5's bar for the device sampler: a CPU implementation (rjmcmc.py pieces + the C oracle's forward
    and Jacobian) driven by the same counter-based streams walks the same chain -- every move, every accept / reject,
    and therefore identical layer-count and interface-depth histograms -- for 4 soundings x 400 iterations."""
    from test_rjmcmc import OracleEngine
    n_it, B = 400, 4
    d, s, dc = _chains(B, 2024, options=priors)
    eo = _emul_options(dc)
    rng = np.random.default_rng(6)
    data = np.tile(d["data"], (B, 1)) * np.r_[1.0, rng.uniform(0.8, 1.3, B - 1)][:, None]
    dc.data.copy_(torch.as_tensor(data))
    dc._initialize()
    sig0 = dc.sigma[:, 0].cpu().numpy()
    eng = OracleEngine("resolve", float(d["z"]))
    chains = []
    for b in range(B):
        sp, vp = _host_priors(dc, b)
        chains.append(rj_emul.Chain(eo, 2024, b, eng, sp, vp, data[b], sig0[b], 0.05, 5.0, dc.n_depth_bins, dc.depth_bin_width))
        assert np.isclose(chains[b].misfit, float(dc.misfit[b]), rtol=1e-9) and np.isclose(chains[b].prior, float(dc.prior[b]), rtol=1e-12)
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
        assert np.array_equal(c.k_hist, dc.k_hist[b].cpu().numpy()) and np.array_equal(c.edge_hist, dc.edge_hist[b].cpu().numpy())
        assert np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-6) and np.allclose(c.sigma, dc.sigma[b, : c.sigma.size].cpu().numpy(), rtol=1e-6)
    assert accs.sum() > 0.2 * accs.size and len(np.unique(ks)) >= 3 and set(np.unique(acts)) == {0, 1, 2, 3}
    assert ks.max() <= dc.K


@pytest.mark.gpu
def test_height_move_on_the_device_equals_cpu_chains_with_the_same_seeds():
    """``solve_z`` on the device (gbp_rj_options.solve_height; the move is pinned to the reference on the host,
    test_rjmcmc.py::test_height_move_reproduces_the_reference_chain): CPU chains with the same counter-based streams -- the stage
    emulation with the C oracle evaluating the remapped model at the current height and the proposal at the proposed one -- walk
    the same chains as the device: every move and decision, the heights to 1e-9 m, the height posterior, over 400 iterations,
    under the fused lock-step driver, the ten-launch driver, concurrent sub-blocks and the persistent kernel (which re-reads the two
    heights and their abscissa windows every iteration);
    cached state equals a from-scratch evaluation at the final heights."""
    from test_rjmcmc import OracleEngine
    n_it, B = 400, 6
    hopt = dict(solve_z=True, maximum_z_change=1.5, z_proposal_variance=0.15)
    runs = {}
    for mode in (1, 3, 2):
        d, s, dc = _chains(B, 2025, options=hopt)
        dc.run_mode = mode
        rng = np.random.default_rng(6)
        data = np.tile(d["data"], (B, 1)) * np.r_[1.0, rng.uniform(0.8, 1.3, B - 1)][:, None]
        dc.data.copy_(torch.as_tensor(data))
        dc._initialize()
        if mode == 1:
            eo = _emul_options(dc)
            sig0 = dc.sigma[:, 0].cpu().numpy()
            eng = OracleEngine("resolve", float(d["z"]))
            chains = []
            for b in range(B):
                sp, vp = _host_priors(dc, b)
                chains.append(rj_emul.Chain(eo, 2025, b, eng, sp, vp, data[b], sig0[b], 0.05, 5.0, dc.n_depth_bins, dc.depth_bin_width,
                                            height=float(d["z"])))
                assert np.isclose(chains[b].prior, float(dc.prior[b]), rtol=1e-12)
            accs, ks, hs = [], [], []
            prev = dc.n_accepted.cpu().numpy().copy()
            for it in range(n_it):
                dc.step()
                now = dc.n_accepted.cpu().numpy()
                accs.append(now - prev); ks.append(dc.k.cpu().numpy().copy()); hs.append(dc.height.cpu().numpy().copy())
                prev = now.copy()
                for c in chains:
                    c.step(it)
                    assert abs(c.height - hs[-1][c.b]) < 1e-9, (it, c.b)
            accs, ks, hs = np.array(accs), np.array(ks), np.array(hs)
            for b, c in enumerate(chains):
                tr = np.array(c.trace)
                assert np.array_equal(tr[:, 1], accs[:, b]) and np.array_equal(tr[:, 2], ks[:, b]), b
                assert np.array_equal(c.height_hist, dc.height_hist[b].cpu().numpy()) and np.array_equal(c.k_hist, dc.k_hist[b].cpu().numpy())
                assert np.isclose(c.misfit, float(dc.misfit[b]), rtol=1e-6)
            assert np.ptp(hs, axis=0).min() > 0.3 and np.all(np.abs(hs - float(d["z"])) <= 1.5) and accs.sum() > 0.2 * accs.size
            assert torch.all(torch.abs(dc.best_height - float(d["z"])) <= 1.5)
        else:
            dc.run(n_it)
        runs[mode] = dc
    for n in ("k", "sigma", "edges", "height", "height_hist", "best_height", "rel", "add", "misfit", "k_hist", "n_accepted"):
        assert torch.equal(getattr(runs[1], n), getattr(runs[3], n)), n
        assert torch.equal(getattr(runs[1], n), getattr(runs[2], n)), n
    # concurrent sub-blocks (a block large enough for them) and the persistent kernel in two rounds, against the one-block driver
    big = {}
    for mode in (1, 4, 2):
        d, s, dc = _chains(2304, 11, options=hopt)
        dc.run_mode = mode
        dc.run(60)
        big[mode] = dc
    for n in ("k", "sigma", "height", "height_hist", "misfit", "n_accepted"):
        assert torch.equal(getattr(big[1], n), getattr(big[4], n)), n
        assert torch.equal(getattr(big[1], n), getattr(big[2], n)), n
    dc = big[4]
    from geobipy_amd import FdemBatch
    thk = torch.zeros_like(dc.sigma)
    thk[:, :-1] = torch.diff(torch.cat([torch.zeros(dc.B, 1, dtype=torch.float64, device=dc.device), dc.edges], dim=1), dim=1)[:, :-1].nan_to_num(posinf=0.0)
    fb = FdemBatch(s, dc.k.cpu().numpy(), dc.sigma.cpu().numpy(), thk.cpu().numpy(), dc.height.cpu().numpy(), hankel_eps_ppm=dc.hankel_eps_ppm)
    assert torch.allclose(fb.forward(), dc.pred, rtol=1e-9, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("exact,kw", [(True, dict()), (False, dict()),
                                      (True, dict(options=dict(maximum_number_of_layers=8, probability_of_birth=0.4))),
                                      (False, dict(options=dict(maximum_number_of_layers=5, probability_of_birth=0.4))),    # (rows shorter than the packed stages' 8-lane group)
                                      (True, dict(reference_schedule=True, burn_in_min_iterations=60, hitmap=True, n_value_bins=20,
                                                  options=dict(n_markov_chains=150)))])
def test_persistent_kernel_walks_the_same_chains(exact, kw):
    """gbp_rj_run_mode: the persistent per-chain kernel (one workgroup owns a chain and loops over the iterations in ONE launch)
    and the lock-step drivers (mode 1: seven launches per iteration, one fused physics launch per stage; mode 3: ten, one
    per kind of evaluation and layer-count bucket; mode 4: mode 1 on concurrent sub-blocks of the chains, each on its own stream)
    are the same device functions on the same arrays, and their results do not
    depend on the wave counts -- the chains, their posteriors and every piece of carried state are bit-identical, also when
    the run is cut into launches of different lengths, for deep models (> 8 layers: the one-wave variants of the per-chain
    algebra and the 8-row-group Jacobian pass) and under the reference's burn-in schedule."""
    B, n_it = 300, 400
    runs = []
    # (mode 4: the block cut into concurrent sub-blocks on streams of their own -- here two of 150 chains -- and a run cut in two)
    for mode, cuts in ((1, (n_it,)), (3, (n_it,)), (2, (n_it,)), (2, (1, 7, 150, n_it - 158)), (4, (n_it,)), (4, (90, n_it - 90))):
        d, s, dc = _chains(B, 31, exact=exact, **{k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()})
        rng = np.random.default_rng(4)
        data = np.tile(d["data"], (B, 1)) * rng.uniform(0.7, 1.4, (B, 1))
        dc.data.copy_(torch.as_tensor(data))
        dc._initialize()
        if "options" not in kw and not kw:                       # plain options: start a third of the chains from deep models
            _load_random_state(dc, np.random.default_rng(12), kmax=dc.K)
            dc.rel.fill_(0.05); dc.add.fill_(5.0)
            thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64)).contiguous()
            dc._eval_loglike(dc.k, dc.sigma, thk, dc.height, dc.data, dc.rel, dc.add, dc.pred, dc.misfit, dc.like)
            dc._eval_jacobian(dc.k, dc.sigma, thk, dc.height, dc.J, dc.K)
            dc.prior.copy_(rg.model_log_prior(dc.edges, dc.sigma, dc.k.to(torch.int64), dc.K, dc.gradient_precision)
                           + rg.log_uniform_prior(dc.rel[:, 0], dc._bounds["rel"][0][0], dc._bounds["rel"][1][0])
                           + rg.log_uniform_prior(dc.add[:, 0], dc._bounds["add"][0][0], dc._bounds["add"][1][0]))
        dc.run_mode = mode
        for n in cuts:
            dc.run(n)
        torch.cuda.synchronize()
        runs.append(dc)
    ref = runs[0]
    assert int(ref.n_accepted.sum()) > 0.01 * B * n_it and len(torch.unique(ref.k)) >= 3
    if not kw:
        assert int((ref.k > 8).sum()) > 5
    names = ["k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "misfit", "n_accepted", "k_hist", "edge_hist", "rel_hist",
             "add_hist", "best_posterior", "best_k", "best_edges", "best_sigma", "log_ratio"]
    if "reference_schedule" in kw:
        names += ["burned_in_iteration", "status", "hitmap"]
        names.remove("log_ratio")                                 # scratch of finished chains: the persistent kernel stops touching them
        assert int((ref.status == 1).sum()) > 0
    for other in runs[1:]:
        for n in names:
            a, b = getattr(ref, n), getattr(other, n)
            assert torch.equal(torch.nan_to_num(a.double(), nan=-1.25), torch.nan_to_num(b.double(), nan=-1.25)), n


@pytest.mark.gpu
def test_fused_accept_and_proposal_across_the_eight_layer_border():
    """in k_rj_step8 a packed 8-lane group and a deep chain's scanning workgroup must never both act on one chain, whatever side
    of the 8-layer border the chain's current and proposed models are on.  Prior-only chains (births and deaths are accepted often) started
    from models of 1 ... 10 layers cross the border hundreds of times; the fused launches (modes 1 and 4 from the second iteration of a
    call) must leave every array as the persistent kernel -- a workgroup per chain, nothing shared -- leaves it, also when the run is cut
    into calls of a few iterations."""
    B, n_it = 2400, 300
    runs, crossings = [], 0
    for mode, cuts in ((2, (n_it,)), (1, (n_it,)), (4, (n_it,)), (4, (3,) + (2,) * 5 + (n_it - 13,)), (1, (2,) * (n_it // 2))):
        d, s, dc = _chains(B, 57, ignore_likelihood=True,
                           options=dict(maximum_number_of_layers=14, probability_of_birth=0.3, probability_of_death=0.3,
                                        probability_of_perturb=0.2, probability_of_no_change=0.2))
        _load_random_state(dc, np.random.default_rng(33), kmax=10)
        dc.rel.fill_(0.05); dc.add.fill_(5.0)
        thk = rg.layer_widths(dc.edges, dc.k.to(torch.int64)).contiguous()
        dc._eval_loglike(dc.k, dc.sigma, thk, dc.height, dc.data, dc.rel, dc.add, dc.pred, dc.misfit, dc.like)
        dc._eval_jacobian(dc.k, dc.sigma, thk, dc.height, dc.J, dc.K)
        dc.prior.copy_(rg.model_log_prior(dc.edges, dc.sigma, dc.k.to(torch.int64), dc.K, dc.gradient_precision)
                       + rg.log_uniform_prior(dc.rel[:, 0], dc._bounds["rel"][0][0], dc._bounds["rel"][1][0])
                       + rg.log_uniform_prior(dc.add[:, 0], dc._bounds["add"][0][0], dc._bounds["add"][1][0]))
        assert float(dc.like.abs().max()) == 0.0                  # (no channel is active: the likelihood of every state is the constant 0)
        dc.run_mode = mode
        k_prev = dc.k.clone()
        for n in cuts:
            dc.run(n)
            if len(cuts) == n_it // 2:                            # (the run in calls of 2 iterations: border crossings between its snapshots)
                crossings += int(((k_prev <= 8) != (dc.k <= 8)).sum())
                k_prev = dc.k.clone()
        torch.cuda.synchronize()
        runs.append(dc)
    ref = runs[0]
    assert crossings > 200, crossings                             # (a lower bound: a chain that crosses twice inside a window is not seen)
    assert int(ref.n_accepted.sum()) > 0.2 * B * n_it
    names = ["k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "misfit", "n_accepted", "k_hist", "edge_hist", "rel_hist",
             "add_hist", "best_posterior", "best_k", "best_edges", "best_sigma", "log_ratio"]
    for other in runs[1:]:
        for n in names:
            a, b = getattr(ref, n), getattr(other, n)
            assert torch.equal(torch.nan_to_num(a.double(), nan=-1.25), torch.nan_to_num(b.double(), nan=-1.25)), n


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
def test_device_chains_sample_like_the_host_chains(exact):
    """The device sampler against the host sampler (rjmcmc.py through BatchedInference -- the code that reproduces
    the reference's chains decision by decision), as ensembles on the reference run's sounding: acceptance rate,
    layer-count distribution and data misfit over iterations 400..800 agree within the ensemble noise."""
    from numpy.random import Generator, PCG64DXSM
    from test_rjmcmc import RESOLVE_OPTIONS
    from geobipy_amd.inference import BatchedInference
    n_burn, n_keep, Bh, Bd = 400, 400, 128, 512
    d, s, dc = _chains(Bd, 5, exact=exact, n_it=n_burn)
    acc0 = dc.n_accepted.clone()
    mis_d, k_d = [], []
    for _ in range(n_keep):
        dc.step()
        mis_d.append(dc.misfit.cpu().numpy())
        k_d.append(dc.k.cpu().numpy())
    acc_d = ((dc.n_accepted - acc0).double() / n_keep).mean().item()
    bi = BatchedInference(s, np.full(Bh, float(d["z"])), np.tile(d["data"], (Bh, 1)),
                          [Generator(PCG64DXSM(900 + b)) for b in range(Bh)], exact_jacobian=exact, **RESOLVE_OPTIONS)
    for _ in range(n_burn):
        bi.step()
    acc_h, mis_h, k_h = [], [], []
    for _ in range(n_keep):
        acc_h.append(bi.step().mean())
        mis_h.append([st.misfit for st in bi.states])
        k_h.append([st.k for st in bi.states])
    acc_h, mis_d, mis_h, k_d, k_h = np.mean(acc_h), np.array(mis_d), np.array(mis_h), np.array(k_d), np.array(k_h)
    hist = lambda k: np.bincount(k.ravel(), minlength=8)[:8] / k.size
    print(f"exact={exact}: acceptance device {acc_d:.3f} host {acc_h:.3f}; mean k {k_d.mean():.3f} / {k_h.mean():.3f}; "
          f"median misfit {np.median(mis_d):.2f} / {np.median(mis_h):.2f}; mean {mis_d.mean():.2f} / {mis_h.mean():.2f}")
    print("k histogram device", np.round(hist(k_d), 3), "host", np.round(hist(k_h), 3))
    assert abs(acc_d - acc_h) < 0.03
    assert abs(k_d.mean() - k_h.mean()) < 0.25
    assert np.abs(hist(k_d) - hist(k_h)).max() < 0.06
    assert abs(np.median(mis_d) - np.median(mis_h)) < 0.1 * np.median(mis_h)


@pytest.mark.gpu
def test_stuck_chains_restart_and_give_up_like_the_reference():
    """infer() with the reference's restart rule (tests/golden/mcmc_reset.npz holds the reference's own runs): chains pinned to
    their starting value by parameter_limits accept nothing, restart after every ``reset_limit`` windows and are given up
    after the fifth restart -- the same update counts as the reference's infer(); chains that move are not touched: their
    run equals the run without the rule, row for row."""
    r = np.load(os.path.join(GOLDEN, "mcmc_reset.npz"))
    B = 24
    for name in ("stuck1", "stuck2", "short"):
        window, reset_limit, n_mc, pinned, failed, updates, n_resets = (int(x) for x in r[name + "_meta"])
        s0 = float(np.sqrt(np.prod(r[name + "_limits"])))     # the fixture's band is 1e-4 wide: narrow enough for the reference's
        d, s, dc = _chains(B, 5, exact=True, reference_schedule=True,                # one chain, not for 24 chains x 100 proposals
                           options=dict(n_markov_chains=n_mc, update_plot_every=window, reset_limit=reset_limit,
                                        parameter_limits=[s0 * (1.0 - 1e-9), s0 * (1.0 + 1e-9)]))
        start = {n: getattr(dc, n).clone() for n in ("sigma", "pred", "prior", "like")}
        assert dc.infer(check_every=7) == B and dc.iteration == updates, name
        assert torch.all(dc.status == 2) and torch.all(dc.n_resets == n_resets) and torch.all(dc.limited == 1)
        assert torch.all(dc.n_accepted == 0) and torch.all(dc.k == 1) and torch.all(dc.iteration0 == updates)
        for n, v in start.items():                                    # back at the initial state after the last restart
            assert torch.equal(getattr(dc, n), v), n
        assert int(dc.k_hist.sum()) == 0
    # chains that move: windows of 20 iterations never pass without an accepted step
    kw = dict(exact=True, reference_schedule=True, burn_in_min_iterations=40)
    wide = dict(n_markov_chains=120, initial_additive_error=400.0, maximum_additive_error=1000.0)   # loose data: high acceptance
    _, _, a = _chains(B, 6, options=dict(wide, update_plot_every=20), **kw)
    _, _, b = _chains(B, 6, options=dict(wide, update_plot_every=1), **kw)                    # 1: rule off (as in the reference)
    assert a.infer(check_every=50) == b.infer(check_every=50)
    assert torch.all(a.n_resets == 0) and torch.all(a.iteration0 == 0)
    for n in ("k", "sigma", "k_hist", "edge_hist", "burned_in_iteration", "status", "best_sigma", "n_accepted"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n


@pytest.mark.gpu
def test_concurrent_sub_blocks_walk_the_same_chains_at_size():
    """Mode 4 at sizes gbp_rj_run picks it for (two concurrent sub-blocks from 2 048 chains; uneven split at 4 099; explicit chain ids):
    every chain, posterior and piece of carried state bit-identical to the one-block lock-step run, and the automatic choice
    (mode 0) takes it."""
    from geobipy_amd import DeviceChains, FdemBatch, FdemSystem, synthetic
    from test_rjmcmc import RESOLVE_OPTIONS
    system = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
    names = ["k", "edges", "sigma", "rel", "add", "pred", "J", "prior", "like", "misfit", "n_accepted", "k_hist", "edge_hist", "rel_hist",
             "add_hist", "best_posterior", "best_k", "best_edges", "best_sigma", "best_rel", "best_add"]
    # (1 802 and 1 950 chains: between the persistent kernel's resident capacity and the three-sub-block threshold -- the automatic choice
    #  runs the first persistently with a short second round and the second as two sub-blocks; the size a survey block shrinks to)
    for B, ids in ((4099, False), (2500, True), (1802, False), (1950, True)):
        nl, sig, thk, h = synthetic.draw_models(B, 4, seed=11)
        data = synthetic.noisy_observations(FdemBatch(system, nl, sig, thk, h, waves=2).forward().cpu().numpy(), seed=12)
        runs = {}
        for mode in (1, 4, 0):
            kw = dict(chain_id=np.arange(B)[::-1].copy() * 3 + 7) if ids else dict(first_chain=1000)
            dc = DeviceChains(system, h, data, seed=5, exact_jacobian=False, hitmap=(mode != 0 and B < 3000), **kw, **o)
            dc.run_mode = mode
            dc.run(150)
            dc.run(37)
            torch.cuda.synchronize()
            runs[mode] = dc
        assert int(runs[1].n_accepted.sum()) > 0.05 * B * 187
        for mode in (4, 0):
            for n in names + (["hitmap"] if (mode != 0 and B < 3000) else []):
                a, b = getattr(runs[1], n), getattr(runs[mode], n)
                assert torch.equal(torch.nan_to_num(a.double(), nan=-1.25), torch.nan_to_num(b.double(), nan=-1.25)), (B, mode, n)


@pytest.mark.gpu
def test_per_iteration_traces_on_the_device():
    """gbp_rj_options.trace_every / gbp_rj_chains.trace_misfit, trace_accept, best_iteration: the reference's per-iteration arrays
    (Inference1D.data_misfit_v[update - 1] = misfit, acceptance_v[update] = accepted, best_iteration; inversion/Inference1D.py:408,
    414, 713, 743, 749) kept on the device at a stride.  Stride 1 = the arrays in full: they add up to the counters the sampler keeps
    anyway and follow the chain step by step; a stride of 3 is exactly every third entry; all drivers write the same traces."""
    B, n_it = 96, 240
    runs = {}
    for key, every, mode in (("full", 1, 1), ("third", 3, 1), ("persistent", 1, 2), ("sub_blocks", 1, 4)):
        d, s, dc = _chains(B if mode != 4 else 2304, 77, trace_every=every, trace_length=-(-(n_it + 10) // every))
        rng = np.random.default_rng(2)
        dc.data.copy_(torch.as_tensor(np.tile(d["data"], (dc.B, 1)) * rng.uniform(0.8, 1.25, (dc.B, 1))))
        dc._initialize()
        dc.run_mode = mode
        if key == "full":                                       # step by step: the trace follows the chain
            mis, acc = [], []
            prev = dc.n_accepted.clone()
            for _ in range(n_it):
                dc.run(1)
                mis.append(dc.misfit.clone()); acc.append((dc.n_accepted - prev).clone()); prev = dc.n_accepted.clone()
            mis, acc = torch.stack(mis, dim=1), torch.stack(acc, dim=1)
            assert torch.equal(dc.trace_misfit[:, :n_it], mis) and torch.isnan(dc.trace_misfit[:, n_it:]).all()
            assert torch.equal(dc.trace_accept[:, 1:n_it + 1].long(), acc) and int(dc.trace_accept[:, 0].sum()) == 0
        else:
            dc.run(n_it)
        torch.cuda.synchronize()
        runs[key] = dc
    full = runs["full"]
    assert torch.equal(full.trace_accept.long().sum(dim=1), full.n_accepted) and int(full.n_accepted.sum()) > 0.05 * B * n_it
    assert bool(((full.best_iteration >= 1) & (full.best_iteration <= n_it)).all())
    third = runs["third"]
    T3 = third.trace_length
    assert torch.equal(torch.nan_to_num(third.trace_misfit, nan=-1.0), torch.nan_to_num(full.trace_misfit[:, ::3][:, :T3], nan=-1.0))
    assert torch.equal(third.trace_accept, full.trace_accept[:, ::3][:, :T3]) and torch.equal(third.best_iteration, full.best_iteration)
    for n in ("trace_misfit", "trace_accept", "best_iteration", "misfit", "k"):
        assert torch.equal(torch.nan_to_num(getattr(runs["persistent"], n).double(), nan=-1.0), torch.nan_to_num(getattr(full, n).double(), nan=-1.0)), n
    # sub-blocks (2 304 chains: three concurrent sub-blocks, every array sliced by rows) against a one-block run of the same chains
    d, s, one = _chains(2304, 77, trace_every=1, trace_length=n_it + 10)
    rng = np.random.default_rng(2)
    one.data.copy_(torch.as_tensor(np.tile(d["data"], (one.B, 1)) * rng.uniform(0.8, 1.25, (one.B, 1))))
    one._initialize()
    one.run_mode = 1
    one.run(n_it)
    for n in ("trace_misfit", "trace_accept", "best_iteration", "n_accepted"):
        assert torch.equal(torch.nan_to_num(getattr(runs["sub_blocks"], n).double(), nan=-1.0), torch.nan_to_num(getattr(one, n).double(), nan=-1.0)), n
