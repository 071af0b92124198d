"""Row 15 of SURVEY 8(a), the caller contract: replay every call the reference's rjMCMC driver
(Inference1D.initialize / accept_reject, 400 iterations of the reference's own resolve_options run, seed from
the options file) made into the hot path -- forward, sensitivity, data_misfit, likelihood(log=True) on
birth / death / perturb proposals with perturbed error levels -- and check every number.
Fixture: tests/golden/mcmc_trace.npz (made by tests/golden/make_mcmc_trace.py from the imported reference)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, LIKE_ATOL, LIKE_RTOL, PRED_ATOL, PRED_RTOL, oracle_system

FORWARD, SENS, MISFIT, LIKE = 0, 1, 2, 3


@pytest.fixture(scope="module")
def trace():
    return np.load(os.path.join(GOLDEN, "mcmc_trace.npz"))


def close(a, b, atol, rtol):
    return np.all(np.abs(np.asarray(a) - np.asarray(b)) <= atol + rtol * np.abs(np.asarray(b)))


def test_trace_shape(trace):
    kind = trace["kind"]
    assert (kind == FORWARD).sum() == 672 and (kind == SENS).sum() == 291
    assert (kind == MISFIT).sum() == 501 and (kind == LIKE).sum() == 401
    assert trace["n_init"] == 204          # 100 half-space trials x (forward + misfit) + the initial model
    assert trace["k"].min() == 1 and trace["k"].max() >= 3 and 0.3 < trace["accepted"].mean() < 0.7


def test_oracle_reproduces_the_callers_calls(trace):
    from oracle import fdem_oracle as fo
    s = oracle_system("resolve")
    kind = trace["kind"]
    for i in np.flatnonzero(kind == FORWARD)[::7]:
        L = int(trace["nl"][i])
        p = fo.predicted_data(s, trace["sigma"][i, :L], trace["thk"][i, :L], trace["z"][i])
        assert close(p, trace["pred"][i], 1e-10, 1e-12)
    for i in np.flatnonzero(kind == SENS)[::5]:
        L = int(trace["nl"][i])
        J = fo.sensitivity(s, trace["sigma"][i, :L], trace["thk"][i, :L], trace["z"][i])
        assert close(np.vstack([J.real, J.imag]), trace["J"][i][:, :L], 1e-9, 1e-9)
    for i in np.flatnonzero((kind == MISFIT) | (kind == LIKE)):
        _, c2, ll, _ = fo.gauss_loglike(trace["pred"][i], trace["data"][i], trace["rel"][i], trace["add"][i])
        if kind[i] == MISFIT:
            assert abs(c2 - trace["chi2"][i]) <= 1e-9 * (1 + abs(trace["chi2"][i]))
        else:
            assert abs(ll - trace["logL"][i]) <= 1e-9 * (1 + abs(trace["logL"][i]))


@pytest.mark.gpu
def test_gpu_replays_the_callers_calls(trace):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    from geobipy_amd import FdemBatch, FdemSystem
    s = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    kind = trace["kind"]
    f = np.flatnonzero(kind == FORWARD)
    b = FdemBatch(s, trace["nl"][f].astype(np.int32), np.where(trace["sigma"][f] > 0, trace["sigma"][f], 1.0),
                  trace["thk"][f], trace["z"][f])
    assert close(b.forward().cpu().numpy(), trace["pred"][f], PRED_ATOL, PRED_RTOL)
    j = np.flatnonzero(kind == SENS)
    b = FdemBatch(s, trace["nl"][j].astype(np.int32), np.where(trace["sigma"][j] > 0, trace["sigma"][j], 1.0),
                  trace["thk"][j], trace["z"][j])
    assert close(b.sensitivity().cpu().numpy(), trace["J"][j], PRED_ATOL, PRED_RTOL)
    m = np.flatnonzero((kind == MISFIT) | (kind == LIKE))
    n = m.size
    b = FdemBatch(s, np.ones(n, dtype=np.int32), np.ones((n, 1)), np.zeros((n, 1)), np.full(n, 30.0),
                  data=trace["data"][m], relative_error=trace["rel"][m], additive_error=trace["add"][m])
    c2, ll = b.loglike(trace["pred"][m])
    c2, ll = c2.cpu().numpy(), ll.cpu().numpy()
    mis, lik = kind[m] == MISFIT, kind[m] == LIKE
    assert close(c2[mis], trace["chi2"][m][mis], LIKE_ATOL, LIKE_RTOL)
    assert close(ll[lik], trace["logL"][m][lik], LIKE_ATOL, LIKE_RTOL)
    # and fused, the way a batched accept_reject would call it: forward + misfit + likelihood of the proposals
    # (every post-initialisation forward is followed by the misfit and, unless rejected early, the likelihood)
    fi = f[f >= trace["n_init"]]
    nxt = fi + 1
    ok = (nxt < kind.size) & (kind[np.minimum(nxt, kind.size - 1)] == MISFIT)
    fi, nxt = fi[ok], nxt[ok]
    fb = FdemBatch(s, trace["nl"][fi].astype(np.int32), np.where(trace["sigma"][fi] > 0, trace["sigma"][fi], 1.0),
                   trace["thk"][fi], trace["z"][fi], data=trace["data"][nxt], relative_error=trace["rel"][nxt],
                   additive_error=trace["add"][nxt])
    c2, _ = fb.forward_loglike()
    assert close(c2.cpu().numpy(), trace["chi2"][nxt], LIKE_ATOL, LIKE_RTOL)


def test_detail_fixture_is_the_same_run(trace):
    """tests/golden/mcmc_detail.npz (per-iteration internals of the same reference run: RNG state, action, remapped
    model, stochastic-Newton H / mean, proposal terms) is the groundwork fixture for SURVEY row f-2; it must
    describe exactly the run whose hot-path calls are replayed above."""
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    assert np.array_equal(d["accepted"], trace["accepted"]) and np.array_equal(d["new_k"], trace["k"])
    assert d["rng_state"].shape == (400, 6) and len(np.unique(d["rng_state"][:, 1])) == 400
    # every accepted proposal's misfit is one of the misfits the caller asked the hot path for
    chi2_calls = trace["chi2"][trace["kind"] == MISFIT]
    acc = np.flatnonzero(d["accepted"])
    assert all(np.any(np.isclose(chi2_calls, d["new_misfit"][i], rtol=1e-14)) for i in acc)
    # structural moves keep the layer count consistent: insert +1, delete -1
    rk, ck, a = d["rem_k"], d["cur_k"], d["action"]
    assert np.all(rk[a == 1] == ck[a == 1] + 1) and np.all(rk[a == 2] == ck[a == 2] - 1)
    assert np.all(rk[(a == 0) | (a == 3)] == ck[(a == 0) | (a == 3)])
