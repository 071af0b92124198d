"""TDEM Jacobian, chi^2 and log-likelihood against REAL gatdaem1d numbers.

gatdaem1d is absent, but the reference's rendered gallery (docs/_sources/examples/Datapoints/plot_{tempest,skytem}_datapoint.rst.txt)
prints its outputs for fully specified inputs; tests/golden/make_tdem_doc_pins.py parsed them into tests/golden/tdem_doc_pins.npz
(data only).  These are the only reference-side numbers for ``gaTdem1dsen`` / ``ga_fm_dlogc`` (TD/tdem1d.py:98-154) and for
``TdemDataPoint.std`` + ``data_misfit`` + ``likelihood`` on real data.  Bars are STATED here and reported per quantity by
``bench.py`` (``tdem.parity``); measured values (oracle / GPU, identical to the digits shown) next to each:

  Tempest, 30 layers, sigma = logspace(-3, 3, 30), J = d pred / d ln sigma  [30 channels x 30 layers], printed with 9 digits
    * |dJ| <= 1.0e-4 max|J|            on every entry                         measured 7.6e-5
    * |dJ| <= 2.0e-3 max|J[row]|       on every entry of every channel        measured 1.3e-3 (last X gate; 3e-5 on the first)
    * median |dJ / J| <= 1.5e-3        over entries >= 1e-3 max|J|            measured 8.2e-4  (99th percentile 1.6e-2)
    * chi^2 72940.71365767403          rel 2e-4                               measured 1.14e-4
    * logL  -36389.6500813217          rel 2e-4                               measured 1.14e-4
    * logL + chi^2 / 2 (the normalisation: N, Sum ln std -- i.e. Tempest_datapoint.std :161-176 itself)   1e-12 rel (measured 1e-15)
    * best half-space 0.01830738 S/m   the SAME grid cell (28 of logspace(-4, 4, 100))
  SkyTEM, sigma = [500, 20] S/m (75 m of 500 S/m: an extreme model -- the prediction is ~ 6 x the data and chi^2 = Sum (pred / 0.05 d)^2)
    * chi^2 643134.8665683016, logL -320327.7331520335     rel 1.2e-2         measured 9.0e-3  (= 4.5e-3 on the prediction)
    * best half-space 0.01047616 S/m   the SAME grid cell (25)
    * "chi^2 of the best half-space" 19656.315144677585: FINDING -- this is NOT the misfit of the best half-space (ours and the
      oracle's: 6311) but of the LAST trial of the search, 1e4 S/m: EmDataPoint.find_best_halfspace (:176-183) leaves
      predictedData at c[-1] and the gallery calls data_misfit() right after.  Pinned as such: rel 2.5e-2, measured 1.9e-2
      (a 1e4 S/m half-space is at the inductive limit: the prediction is ~ 2 % of the data, chi^2 ~ Sum (1 - p/d)^2 / 0.05^2, so
      1.9e-2 is an ABSOLUTE 1e-2 d on p/d ... the dB/dt response of a near-perfect conductor is a small difference of large terms).
The forward bars of tests/test_tdem.py (1e-3 / 1e-2 on every gate) stand; nothing here needed widening.  The reference's own
criterion (np.allclose rtol 1e-5, tests/test_synthetic_data.py:48, 65) is met for FDEM, vacuous for SkyTEM (atol 1e-8 >> 1e-11
data) and NOT met for Tempest -- see DESIGN.md section 5.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

J_ATOL_ALL, J_ATOL_ROW, J_MEDIAN_REL = 1.0e-4, 2.0e-3, 1.5e-3
TEMPEST_LIKE_RTOL, SKYTEM_LIKE_RTOL, SKYTEM_LAST_TRIAL_RTOL = 2.0e-4, 1.2e-2, 2.5e-2


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(GOLDEN, "tdem_doc_pins.npz"))


def jacobian_report(J, Jr):
    d = np.abs(J - Jr)
    big = np.abs(Jr) >= 1e-3 * np.abs(Jr).max()
    return {"max_abs_over_max": float(d.max() / np.abs(Jr).max()),
            "max_abs_over_row_max": float((d.max(axis=1) / np.abs(Jr).max(axis=1)).max()),
            "median_rel": float(np.median((d / np.abs(Jr))[big]))}


def assert_jacobian(J, Jr):
    r = jacobian_report(J, Jr)
    assert r["max_abs_over_max"] <= J_ATOL_ALL and r["max_abs_over_row_max"] <= J_ATOL_ROW and r["median_rel"] <= J_MEDIAN_REL, r
    return r


def tempest_std(p):
    data = p["tempest_secondary"] + np.repeat(p["tempest_primary"], 15)          # Tempest_datapoint.data :106-113
    return data, np.sqrt((np.repeat(p["tempest_relative_error"], 15) * data) ** 2 + p["tempest_additive_error"] ** 2)   # .std :161-176


def skytem_std(p, off_times):
    add = np.repeat(p["skytem_additive_error"], [26, 19]) * np.sqrt(1e-3 / off_times)     # TdemDataPoint.std :361-365
    return np.sqrt((np.repeat(p["skytem_relative_error"], [26, 19]) * p["skytem_data"]) ** 2 + add ** 2)


def gauss(pred, data, std):
    chi2 = np.sum(((pred - data) / std) ** 2)
    return chi2, -0.5 * data.size * np.log(2 * np.pi) - np.sum(np.log(std)) - 0.5 * chi2


def test_the_pins_are_self_consistent(pins):
    """sensitivity() and fm_dlogc() printed the same matrix; the printed logL and chi^2 differ by the Gaussian normalisation of the
    std the example set -- which pins Tempest_datapoint.std (relative level x TOTAL field, per-channel additive level) to 1e-12
    without any forward model."""
    p = pins
    assert np.array_equal(p["tempest_J"], p["tempest_J_fm_dlogc"])
    _, std = tempest_std(p)
    const = -15.0 * np.log(2 * np.pi) - np.sum(np.log(std))
    assert abs(p["tempest_logl"] + 0.5 * p["tempest_chi2"] - const) <= 1e-12 * abs(const)
    # the gallery's best half-spaces are cells of logspace(-4, 4, 100) (printed with 8 digits)
    c = np.logspace(-4, 4, 100)
    assert abs(c[28] / p["tempest_best_halfspace"] - 1) < 1e-6 and abs(c[25] / p["skytem_best_halfspace"] - 1) < 1e-6


def test_oracle_tempest_jacobian_and_likelihood_vs_gatdaem1d(pins):
    """CPU tier: oracle/tdem_oracle.py (central differences of its forward in ln sigma) against gatdaem1d's printed Jacobian,
    chi^2 and logL."""
    from oracle import tdem_oracle as to
    p = pins
    te = to.parse_stm(os.path.join(GOLDEN, "tempest.stm"))
    sig, thk, g = p["tempest_sigma"], np.diff(np.r_[0.0, p["tempest_edges"]]), p["tempest_geometry"]
    f = lambda s: to.forward(te, s, thk, g[0], g[4], g[5], g[6])
    J, h = np.empty((30, 30)), 1e-4
    for k in range(30):
        sp, sm = sig.copy(), sig.copy()
        sp[k] *= np.exp(h)
        sm[k] *= np.exp(-h)
        J[:, k] = (f(sp) - f(sm)) / (2 * h)
    assert_jacobian(J, p["tempest_J"])
    data, std = tempest_std(p)
    chi2, logl = gauss(f(sig) + np.repeat(p["tempest_primary"], 15), data, std)
    assert abs(chi2 / p["tempest_chi2"] - 1) <= TEMPEST_LIKE_RTOL and abs(logl / p["tempest_logl"] - 1) <= TEMPEST_LIKE_RTOL


def test_oracle_skytem_likelihood_and_half_spaces_vs_gatdaem1d(pins):
    from oracle import tdem_oracle as to
    p = pins
    hm, lm = (to.parse_stm(os.path.join(GOLDEN, n)) for n in ("SkytemHM.stm", "SkytemLM.stm"))
    g, d = p["skytem_geometry"], p["skytem_data"]
    f = lambda s, t: np.r_[to.forward(hm, s, t, g[0], g[4], g[5], g[6]), to.forward(lm, s, t, g[0], g[4], g[5], g[6])]
    std = skytem_std(p, np.r_[hm["windows"].mean(axis=1), lm["windows"].mean(axis=1)])
    chi2, logl = gauss(f(p["skytem_sigma"], [75.0]), d, std)
    assert abs(chi2 / p["skytem_chi2"] - 1) <= SKYTEM_LIKE_RTOL and abs(logl / p["skytem_logl"] - 1) <= SKYTEM_LIKE_RTOL
    c = np.logspace(-4, 4, 100)
    sub = [0, 20, 23, 24, 25, 26, 27, 30, 60, 99]               # (the full 100-cell search runs in the GPU tier)
    phi = np.array([gauss(f([c[i]], []), d, std)[0] for i in sub])
    assert sub[int(np.argmin(phi))] == 25
    assert abs(phi[-1] / p["skytem_chi2_best_halfspace"] - 1) <= SKYTEM_LAST_TRIAL_RTOL     # the LAST trial, see the module docstring
    assert phi[4] < 0.5 * p["skytem_chi2_best_halfspace"]                                   # ... not the best one


# ------------------------------------------------------------------------------------------------------------------------ GPU
def _tempest_point(p):
    from geobipy_amd import CircularLoop, TempestDataPoint
    g = p["tempest_geometry"]
    tx = CircularLoop(x=[0.0], y=[0.0], z=[g[0]], orientation=["z"], radius=[1.0])
    rx = CircularLoop(x=[g[4]], y=[g[5]], z=[g[0] + g[6]], orientation=["x"], radius=[1.0])
    dp = TempestDataPoint(z=float(g[0]), system=[os.path.join(GOLDEN, "tempest.stm")], transmitter_loop=tx, receiver_loop=rx,
                          secondary_field=p["tempest_secondary"], primary_field=p["tempest_primary"],
                          channel_additive_error=p["tempest_additive_error"])
    dp.relative_error = p["tempest_relative_error"]
    return dp


def _skytem_point(p):
    from geobipy_amd import CircularLoop, TdemDataPoint
    g = p["skytem_geometry"]
    tx = CircularLoop(x=[0.0], y=[0.0], z=[g[0]], orientation=["z"], radius=[10.416])
    rx = CircularLoop(x=[g[4]], y=[g[5]], z=[g[0] + g[6]], orientation=["z"], radius=[10.416])
    dp = TdemDataPoint(z=float(g[0]), data=p["skytem_data"], system=[os.path.join(GOLDEN, "SkytemHM.stm"), os.path.join(GOLDEN, "SkytemLM.stm")],
                       transmitter_loop=tx, receiver_loop=rx)
    dp.relative_error, dp.additive_error = p["skytem_relative_error"], p["skytem_additive_error"]
    return dp


def gpu_report(p):
    """Every pinned quantity from the GPU path, as relative errors -- used by the tests below and by bench.py (``tdem.parity``)."""
    from geobipy_amd import Model, RectilinearMesh1D
    from geobipy_amd.tdem import NativeTdemSystem, TdemBatch, TdemSystem
    out = {}
    # (1) batched Python host over the kernels, (2) C-level entry gbp_tdem_fm_dlogc, (3) the object API -- the same Jacobian three ways
    sig, thk = p["tempest_sigma"][None, :], np.r_[np.diff(np.r_[0.0, p["tempest_edges"]]), 0.0][None, :]
    g = p["tempest_geometry"]
    b = TdemBatch(TdemSystem(os.path.join(GOLDEN, "tempest.stm")), np.array([30]), sig, thk, np.array([g[0]]), tuple(g[4:7]))
    pred, J = b.fm_dlogc()
    out["J_batch"], out["pred_batch"] = J.cpu().numpy()[0], pred.cpu().numpy()[0]
    c = NativeTdemSystem(os.path.join(GOLDEN, "tempest.stm"))
    pc, Jc = c.fm_dlogc(g[None, :], np.array([30]), sig, thk)
    out["J_c"], out["pred_c"] = Jc.cpu().numpy()[0], pc.cpu().numpy()[0]
    dp = _tempest_point(p)
    mod = Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, p["tempest_edges"], np.inf]), values=p["tempest_sigma"])
    out["J_object"] = np.array(dp.sensitivity(mod))
    dp.fm_dlogc(mod)
    out["J_object_fm_dlogc"] = np.array(dp.sensitivity_matrix)
    out["tempest_chi2"], out["tempest_logl"] = float(dp.data_misfit()), float(dp.likelihood(log=True))
    out["tempest_best_halfspace"] = float(dp.find_best_halfspace().values[0])
    sk = _skytem_point(p)
    sk.forward(Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, p["skytem_edges"], np.inf]), values=p["skytem_sigma"]))
    out["skytem_chi2"], out["skytem_logl"] = float(sk.data_misfit()), float(sk.likelihood(log=True))
    hs = sk.find_best_halfspace()
    out["skytem_best_halfspace"] = float(hs.values[0])
    out["skytem_chi2_after_search"] = float(sk.data_misfit())           # the LAST trial's, like the reference (module docstring)
    sk.forward(hs)
    out["skytem_chi2_best_halfspace"] = float(sk.data_misfit())
    return out


@pytest.mark.gpu
def test_gpu_tdem_jacobian_and_likelihood_vs_gatdaem1d(pins):
    """-m gpu: TdemBatch.fm_dlogc, gbp_tdem_fm_dlogc (C level), TempestDataPoint.sensitivity / fm_dlogc / data_misfit / likelihood /
    find_best_halfspace and the SkyTEM twins against gatdaem1d's printed numbers."""
    pytest.importorskip("torch")
    p = pins
    r = gpu_report(p)
    for k in ("J_batch", "J_c", "J_object", "J_object_fm_dlogc"):
        assert r[k].shape == (30, 30)
        assert_jacobian(r[k], p["tempest_J"])
    assert np.abs(r["J_batch"] - r["J_c"]).max() <= 1e-9 * np.abs(r["J_batch"]).max()
    assert np.abs(r["J_batch"] - r["J_object"]).max() <= 1e-9 * np.abs(r["J_batch"]).max()
    assert abs(r["tempest_chi2"] / p["tempest_chi2"] - 1) <= TEMPEST_LIKE_RTOL
    assert abs(r["tempest_logl"] / p["tempest_logl"] - 1) <= TEMPEST_LIKE_RTOL
    assert abs(r["tempest_best_halfspace"] / p["tempest_best_halfspace"] - 1) < 1e-6
    assert abs(r["skytem_chi2"] / p["skytem_chi2"] - 1) <= SKYTEM_LIKE_RTOL
    assert abs(r["skytem_logl"] / p["skytem_logl"] - 1) <= SKYTEM_LIKE_RTOL
    assert abs(r["skytem_best_halfspace"] / p["skytem_best_halfspace"] - 1) < 1e-6
    assert abs(r["skytem_chi2_after_search"] / p["skytem_chi2_best_halfspace"] - 1) <= SKYTEM_LAST_TRIAL_RTOL
    assert r["skytem_chi2_best_halfspace"] < 0.5 * p["skytem_chi2_best_halfspace"]
    # the GPU path and the oracle agree far inside these bars (same pipeline, independent code)
    from oracle import tdem_oracle as to
    te = to.parse_stm(os.path.join(GOLDEN, "tempest.stm"))
    g = p["tempest_geometry"]
    o = to.forward(te, p["tempest_sigma"], np.diff(np.r_[0.0, p["tempest_edges"]]), g[0], g[4], g[5], g[6])
    assert np.abs(r["pred_batch"] - o).max() <= 1e-8 * np.abs(o).max() and np.abs(r["pred_c"] - o).max() <= 1e-8 * np.abs(o).max()
