"""GPU tier (-m gpu): parity of the HIP path, called through the C ABI, against the oracle and the fixtures.

Tolerances (SURVEY section 7 hard part 1):
    predicted data  |d| <= 1e-7 ppm + 1e-9 |ref|     chi^2, logL  |d| <= 1e-6 + 1e-9 |ref|
/root/reference is never touched here: fixtures come from tests/golden, the checker is oracle/.
"""
import os

import numpy as np
import pytest

from conftest import (GOLDEN, LIKE_ATOL, LIKE_RTOL, PRED_ATOL, PRED_RTOL, WEDGE_CONDUCTIVITY, oracle_system,
                      read_clean_csv, wedge_models)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from geobipy_amd import _lib
    _lib.load()      # the native library must be the thing that runs: fail loudly if absent


def close(a, b, atol, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def product_system(name):
    from geobipy_amd import FdemSystem
    return FdemSystem.read(os.path.join(GOLDEN, f"{name}.stm"))


def pad(a, Lmax, fill):
    out = np.full((a.shape[0], Lmax), fill, dtype=np.float64)
    out[:, : a.shape[1]] = a
    return out


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_fixtures_of_imported_reference(golden_npz, name):
    """forward, chi^2, logL for every fixture (all tensor ids, L = 1..30, NaN channels) in ONE ragged batch."""
    from geobipy_amd import FdemBatch
    g, s = golden_npz, product_system(name)
    Lmax = 32
    nl, sig, thk, h, obs, rel, add, pred, chi2, logl = [], [], [], [], [], [], [], [], [], []
    for L in [1, 2, 3, 5, 8, 30]:
        k = f"{name}_L{L}"
        n = g[k + "/sigma"].shape[0]
        t = g[k + "/thk"].copy()
        t[:, -1] = 0.0
        nl += [L] * n
        sig.append(pad(g[k + "/sigma"], Lmax, 1.0))
        thk.append(pad(t, Lmax, 0.0))
        for lst, key in [(h, "height"), (obs, "obs"), (rel, "rel"), (add, "add"), (pred, "pred"), (chi2, "chi2"),
                         (logl, "logL")]:
            lst.append(g[f"{k}/{key}"])
    cat = np.concatenate
    b = FdemBatch(s, np.array(nl), cat(sig), cat(thk), cat(h), data=cat(obs), relative_error=cat(rel),
                  additive_error=cat(add))
    b.validate()
    c2, ll = b.forward_loglike()
    torch.cuda.synchronize()
    assert close(b.predicted.cpu().numpy(), cat(pred), PRED_ATOL, PRED_RTOL)
    assert close(c2.cpu().numpy(), cat(chi2), LIKE_ATOL, LIKE_RTOL)
    assert close(ll.cpu().numpy(), cat(logl), LIKE_ATOL, LIKE_RTOL)
    # un-fused entry points give the same numbers
    p2 = b.forward(out=torch.empty_like(b.predicted)).cpu().numpy()
    assert np.array_equal(p2, b.predicted.cpu().numpy())
    c3, l3 = b.loglike(torch.as_tensor(cat(pred)))
    assert close(c3.cpu().numpy(), cat(chi2), LIKE_ATOL, LIKE_RTOL)
    assert close(l3.cpu().numpy(), cat(logl), LIKE_ATOL, LIKE_RTOL)


@pytest.mark.parametrize("model_type", sorted(WEDGE_CONDUCTIVITY))
def test_reference_known_answer_files(model_type):
    """The reference's own test (tests/test_synthetic_data.py:16-30): 79-sounding wedge vs resolve_*_clean.csv."""
    from geobipy_amd import FdemBatch
    s = product_system("resolve")
    csv = read_clean_csv(os.path.join(GOLDEN, f"resolve_{model_type}_clean.csv"))
    thk = wedge_models()
    sig = np.tile(np.asarray(WEDGE_CONDUCTIVITY[model_type]), (79, 1))
    p = FdemBatch(s, np.full(79, 3), sig, thk, np.full(79, 30.0)).forward().cpu().numpy()
    assert np.allclose(p, csv)                                   # the reference's criterion
    assert close(p, csv, PRED_ATOL, PRED_RTOL)                   # ours


def test_datapoint_interface_config1():
    """BASELINE config 1 through the reference-style objects (FdemDataPoint / Model / RectilinearMesh1D)."""
    from geobipy_amd import FdemDataPoint, Model, RectilinearMesh1D
    csv0 = read_clean_csv(os.path.join(GOLDEN, "resolve_glacial_clean.csv"))[0]
    mod = Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, 5.0, 7.5, np.inf]), values=np.r_[1e-2, 1e-1, 0.03333333])
    dp = FdemDataPoint(x=0.0, y=0.0, z=30.0, elevation=0.0, data=1.03 * csv0, std=None,
                       system=os.path.join(GOLDEN, "resolve.stm"))
    dp.relative_error = 0.05
    dp.additive_error = 5.0
    dp.forward(mod)
    ref = np.array([41.08919660004556, 225.2575637837353, 151.27195180787993, 837.5752928944062,
                    1922.7662024927201, 2513.018568364122, 136.76679901024212, 406.1487690660227,
                    209.11121320604505, 807.6012738287338, 869.5433434122464, 649.9213242250147])
    assert close(dp.predictedData, ref, PRED_ATOL, PRED_RTOL)
    assert abs(dp.data_misfit() - 3.4175602330457497) <= LIKE_ATOL
    assert abs(dp.likelihood(log=True) - (-51.18449066872814)) <= LIKE_ATOL
    assert np.isclose(dp.likelihood(log=False), np.exp(-51.18449066872814), rtol=1e-6)
    # the fused results cached by forward() are dropped as soon as anything they depend on changes
    from oracle import fdem_oracle as fo
    dp.relative_error = 0.08
    _, c2, ll, _ = fo.gauss_loglike(dp.predictedData, dp.data, 0.08, 5.0)
    assert abs(dp.data_misfit() - c2) <= LIKE_ATOL and abs(dp.likelihood(log=True) - ll) <= LIKE_ATOL
    dp.predictedData[2] += 1.0
    _, c2, ll, _ = fo.gauss_loglike(dp.predictedData, dp.data, 0.08, 5.0)
    assert abs(dp.data_misfit() - c2) <= LIKE_ATOL and abs(dp.likelihood(log=True) - ll) <= LIKE_ATOL
    import copy
    dp2 = copy.deepcopy(dp)                       # Inference1D deep-copies the datapoint every iteration
    dp2.forward(mod)
    assert close(dp2.predictedData, ref, PRED_ATOL, PRED_RTOL) and dp2.system is dp.system
    with pytest.raises(AssertionError):      # last edge must be infinite (FdemDataPoint.py:541)
        dp.forward(Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, 5.0, 7.5]), values=np.r_[1e-2, 1e-1]))


@pytest.mark.parametrize("B,L", [(4096, 5), (1000, 8), (37, 1), (1, 3)])
def test_random_batches_vs_oracle(B, L):
    """BASELINE config 2 shape (4096 x 10 freq x 5 layers) and friends against the oracle on identical inputs."""
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=100 + B)
    clean = FdemBatch(s, nl, sig, thk, h).forward().cpu().numpy()
    obs = synthetic.noisy_observations(clean, seed=200 + B)
    obs[::7, 3] = np.nan
    prop = synthetic.redraw_sigma(B, L, seed=300 + B)
    rel, add = np.full(B, 0.05), np.full(B, 5.0)
    b = FdemBatch(s, nl, prop, thk, h, data=obs, relative_error=rel, additive_error=add)
    c2, ll = b.forward_loglike()
    torch.cuda.synchronize()
    n = min(B, 512)          # the oracle finishes these in seconds
    p_ref, c_ref, l_ref = fo.forward_loglike_batch(oracle_system("syn10"), nl[:n], prop[:n], thk[:n], h[:n], obs[:n],
                                                   rel[:n], add[:n], nthreads=0)
    assert close(b.predicted[:n].cpu().numpy(), p_ref, PRED_ATOL, PRED_RTOL)
    assert close(c2[:n].cpu().numpy(), c_ref, LIKE_ATOL, LIKE_RTOL)
    assert close(ll[:n].cpu().numpy(), l_ref, LIKE_ATOL, LIKE_RTOL)
    assert torch.isfinite(b.predicted).all() and torch.isfinite(c2).all() and torch.isfinite(ll).all()


def test_ragged_layers_padding_and_order_independence():
    """Ragged batch (1..30 layers): padding columns are never read and results do not depend on batch order."""
    from geobipy_amd import FdemBatch, synthetic
    rng = np.random.default_rng(5)
    s = product_system("resolve")
    B, Lmax = 600, 30
    nl = rng.integers(1, Lmax + 1, size=B).astype(np.int32)
    _, sig, thk, h = synthetic.draw_models(B, Lmax, seed=11)
    p1 = FdemBatch(s, nl, sig, thk, h).forward().cpu().numpy()
    sig2, thk2 = sig.copy(), thk.copy()
    for i in range(B):
        sig2[i, nl[i]:] = np.nan          # poison everything the kernel must not read
        thk2[i, nl[i] - 1:] = np.nan
    p2 = FdemBatch(s, nl, sig2, thk2, h).forward().cpu().numpy()
    assert np.array_equal(p1, p2) and np.isfinite(p1).all()
    perm = rng.permutation(B)
    p3 = FdemBatch(s, nl[perm], sig[perm], thk[perm], h[perm]).forward().cpu().numpy()
    assert np.array_equal(p3, p1[perm])
    # wider row stride, same answer
    p4 = FdemBatch(s, nl, pad(sig, 40, 7.0), pad(thk, 40, 3.0), h).forward().cpu().numpy()
    assert np.array_equal(p4, p1)


def test_both_branches_of_the_complex_square_root_in_one_launch():
    """Soundings whose layers all conduct above 4 omega_max eps0 (2.9e-5 S/m at 130 kHz) take the select-free complex
    square root, the others the general one (the kernel decides per sounding); both against the oracle, including
    conductivities right at the threshold and layers where displacement currents dominate."""
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    B, L = 600, 6
    rng = np.random.default_rng(17)
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=41)
    thr = 4.0 * 2.0 * np.pi * 129550.0 * 8.8541878128e-12
    sig[np.arange(200), rng.integers(0, L, 200)] = np.exp(rng.uniform(np.log(1e-7), np.log(2e-5), 200))   # general branch
    sig[200:300, 2] = thr * rng.uniform(0.98, 1.02, 100)                                             # at the threshold
    sig[300:400] = np.exp(rng.uniform(np.log(3e-5), np.log(3e-4), (100, L)))                         # direct branch, poor conductors
    h[::3] = rng.uniform(0.5, 5.0, h[::3].size)
    b = FdemBatch(s, nl, sig, thk, h)
    p = b.forward().cpu().numpy()
    J = b.sensitivity().cpu().numpy()
    p_ref, _, _ = fo.forward_loglike_batch(oracle_system("syn10"), nl, sig, thk, h, np.ones_like(p), np.full(B, 0.05), np.full(B, 5.0),
                                           nthreads=0)
    assert np.isfinite(p).all() and close(p, p_ref, PRED_ATOL, PRED_RTOL)
    for i in (0, 150, 250, 350, 500):
        Jo = fo.sensitivity(oracle_system("syn10"), sig[i], thk[i], h[i])
        assert close(J[i], np.vstack([Jo.real, Jo.imag]), PRED_ATOL, 10 * PRED_RTOL), i


def test_full_size_properties():
    """BASELINE full size (65 536 x 10 freq x 8 layers): size-independent properties instead of the oracle.
    (a) splitting a layer in two with equal conductivity leaves the response unchanged;
    (b) a layered model with all-equal conductivities equals the half-space;
    (c) chi^2 / logL from the fused kernel equal the stand-alone likelihood kernel on the same predictions;
    (d) a spot sample agrees with the oracle."""
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    B, L = 65536, 8
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=42)
    base = FdemBatch(s, nl, sig, thk, h)
    p = base.forward().clone()
    # (a) split layer 2 (0-based) into two halves -> 9 layers
    sig9 = np.concatenate([sig[:, :3], sig[:, 2:]], axis=1)
    thk9 = np.concatenate([thk[:, :2], 0.5 * thk[:, 2:3], 0.5 * thk[:, 2:3], thk[:, 3:]], axis=1)
    p9 = FdemBatch(s, np.full(B, 9), sig9, thk9, h).forward()
    assert torch.all((p9 - p).abs() <= PRED_ATOL + PRED_RTOL * p.abs())
    # (b) uniform conductivity == half-space
    sigu = np.repeat(sig[:, :1], L, axis=1)
    pu = FdemBatch(s, nl, sigu, thk, h).forward().clone()
    ph = FdemBatch(s, np.ones(B, dtype=np.int32), sig[:, :1].copy(), np.zeros((B, 1)), h).forward()
    assert torch.all((pu - ph).abs() <= PRED_ATOL + PRED_RTOL * ph.abs())
    # (c) fused vs stand-alone likelihood
    obs = synthetic.noisy_observations(p.cpu().numpy(), seed=43)
    fb = FdemBatch(s, nl, synthetic.redraw_sigma(B, L, seed=44), thk, h, data=obs, relative_error=np.full(B, 0.05),
                   additive_error=np.full(B, 5.0))
    c2, ll = fb.forward_loglike()
    c2, ll = c2.clone(), ll.clone()
    c2b, llb = fb.loglike(fb.predicted)
    assert torch.all((c2 - c2b).abs() <= 1e-9 * (1 + c2b.abs())) and torch.all((ll - llb).abs() <= 1e-9 * (1 + llb.abs()))
    assert torch.isfinite(c2).all() and torch.isfinite(ll).all()
    # (d) spot sample vs oracle
    idx = np.arange(0, B, B // 256)
    p_ref, c_ref, l_ref = fo.forward_loglike_batch(oracle_system("syn10"), nl[idx], fb.sigma.cpu().numpy()[idx],
                                                   thk[idx], h[idx], obs[idx], np.full(idx.size, 0.05),
                                                   np.full(idx.size, 5.0), nthreads=0)
    assert close(fb.predicted.cpu().numpy()[idx], p_ref, PRED_ATOL, PRED_RTOL)
    assert close(c2.cpu().numpy()[idx], c_ref, LIKE_ATOL, LIKE_RTOL)
    assert close(ll.cpu().numpy()[idx], l_ref, LIKE_ATOL, LIKE_RTOL)


def test_edge_cases():
    from geobipy_amd import FdemBatch, _lib, synthetic
    from geobipy_amd.system import CircularLoop, FdemSystem
    s = synthetic.syn10_system()
    # empty batch: legal, no launch
    nl, sig, thk, h = synthetic.draw_models(0, 3)
    b = FdemBatch(s, nl, sig.reshape(0, 3), thk.reshape(0, 3), h, data=np.zeros((0, 20)), relative_error=np.zeros(0),
                  additive_error=np.zeros(0))
    assert b.forward().shape == (0, 20)
    assert b.forward_loglike()[0].shape == (0,)
    # all channels inactive: chi2 = 0, logL = 0 (N_a = 0)
    nl, sig, thk, h = synthetic.draw_models(3, 4, seed=1)
    b = FdemBatch(s, nl, sig, thk, h, data=-np.ones((3, 20)), relative_error=np.full(3, 0.05),
                  additive_error=np.full(3, 5.0))
    c2, ll = b.forward_loglike()
    assert torch.all(c2 == 0) and torch.all(ll == 0)
    # unsupported tensor id (Tx y): refused at system creation, like the oracle
    bad = FdemSystem([1000.0], CircularLoop(orientation=["y"], moment=[1.0], x=[0.0], y=[0.0], z=[0.0]),
                     CircularLoop(orientation=["z"], moment=[1.0], x=[8.0], y=[0.0], z=[0.0]))
    with pytest.raises(_lib.NativeLibraryError):
        bad.handle()
    # very thick / very conductive layers (exp underflow path) and very resistive thin ones stay finite
    nl = np.array([3, 3], dtype=np.int32)
    sig = np.array([[10.0, 5.0, 1.0], [1e-6, 1e-6, 1e-6]])
    thk = np.array([[500.0, 800.0, 0.0], [0.01, 0.01, 0.0]])
    p = FdemBatch(s, nl, sig, thk, np.array([30.0, 30.0])).forward()
    assert torch.isfinite(p).all()
    from oracle import fdem_oracle as fo
    ref = np.stack([fo.predicted_data(oracle_system("syn10"), sig[i], thk[i], 30.0) for i in range(2)])
    assert close(p.cpu().numpy(), ref, PRED_ATOL, PRED_RTOL)


def test_bad_rows_are_flagged_and_the_rest_of_the_batch_is_untouched():
    """SURVEY 8b "Errors": a row with more layers than the launch was sized for (nlayers > Lmax, or > max_layers of the
    Jacobian entries) comes back as NaN -- no overrun of the LDS layer tables or of its neighbours' rows --, a row with
    nlayers = 0 is skipped, a non-positive conductivity poisons its own row only; gbp_fdem_validate names the cause."""
    from geobipy_amd import FdemBatch, _lib, synthetic
    s = synthetic.syn10_system()
    B, L, Lmax = 64, 5, 8
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=21, Lmax=Lmax)
    good = FdemBatch(s, nl, sig, thk, h, data=np.full((B, 20), 100.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
    p0 = good.forward().clone()
    c0, l0 = (t.clone() for t in good.forward_loglike())
    J0 = good.fm_dlogc(exact=True).clone()
    nl2, sig2, h2 = nl.copy(), sig.copy(), h.copy()
    nl2[3], nl2[17], nl2[40] = Lmax + 5, 1000000, 0          # too many layers (twice), skipped row
    sig2[9, 2] = -1.0                                            # bad conductivity
    h2[11] = -3.0                                                # sensor below the surface
    bad = FdemBatch(s, nl2, sig2, thk, h2, data=np.full((B, 20), 100.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
    bad.predicted.fill_(-7.0); bad.chi2.fill_(-7.0); bad.logL.fill_(-7.0)
    c1, l1 = bad.forward_loglike()
    p1 = bad.predicted
    ok = np.ones(B, bool); ok[[3, 17, 40, 9, 11]] = False
    okt = torch.as_tensor(ok, device=p1.device)
    assert torch.equal(p1[okt], p0[okt]) and torch.equal(c1[okt], c0[okt]) and torch.equal(l1[okt], l0[okt])
    assert torch.isnan(p1[[3, 17]]).all() and torch.isnan(c1[[3, 17]]).all() and torch.isnan(l1[[3, 17]]).all()
    assert (p1[40] == -7.0).all() and c1[40] == -7.0                                  # skipped: nothing written
    st = bad.status().cpu().numpy()
    assert st[3] == 1 | 16 and st[17] == 1 | 16 and st[40] == 1 and st[9] & 2 and st[11] & 8 and np.all(st[ok] == 0)
    assert np.all(good.status().cpu().numpy() == 0)
    good.validate()
    with pytest.raises(AssertionError):
        bad.validate()
    # Jacobian entries: max_layers smaller than a row's layer count
    J1 = torch.full((B, 20, Lmax), -7.0, dtype=torch.float64, device=p1.device)
    pj = torch.full((B, 20), -7.0, dtype=torch.float64, device=p1.device)
    nlj = nl.copy(); nlj[5] = 7                                  # a legal 7-layer row, but the launch is sized for 5
    nlj_t = torch.as_tensor(nlj, dtype=torch.int32, device=p1.device)
    _lib.check(_lib.load().gbp_fdem_fm_dlogc(good._h.ptr, B, Lmax, nlj_t.data_ptr(), good.sigma.data_ptr(), good.thk.data_ptr(),
                                             good.height.data_ptr(), pj.data_ptr(), J1.data_ptr(), 5, 1, None))
    torch.cuda.synchronize()
    keep = np.ones(B, bool); keep[5] = False
    kt = torch.as_tensor(keep, device=p1.device)
    assert torch.isnan(J1[5]).all() and torch.isnan(pj[5]).all() and torch.equal(J1[kt], J0[kt])


def test_results_do_not_depend_on_the_waves_per_sounding():
    """gbp_fdem_forward_ex / _loglike_ex: the waves per workgroup are an explicit argument (no hidden per-thread state) and a
    performance hint only -- every 64-point pass reduces to one partial sum per frequency and a frequency's partials are added
    in pass order, so 1, 2, 3, 4, 8 or 16 waves, a large batch (one wave per sounding chosen automatically) or a small one (many
    waves) return the same bits.  Default abscissa windows and all abscissae (passes that straddle two frequencies)."""
    from geobipy_amd import FdemBatch, synthetic
    s = synthetic.syn10_system()
    nl, sig, thk, h = synthetic.draw_models(70000, 8, seed=33)
    data, rel, add = np.full((70000, 20), 80.0), np.full(70000, 0.05), np.full(70000, 5.0)
    for eps in (None, 0.0):
        ref = FdemBatch(s, nl, sig, thk, h, data=data, relative_error=rel, additive_error=add, hankel_eps_ppm=eps)     # waves = 0: 1 wave at this size
        rc, rl = (t.clone() for t in ref.forward_loglike())
        for w in (1, 2, 3, 4, 8, 16, 0):
            n = 70000 if w in (1, 4) else 3000
            fb = FdemBatch(s, nl[:n], sig[:n], thk[:n], h[:n], data=data[:n], relative_error=rel[:n], additive_error=add[:n], waves=w,
                           hankel_eps_ppm=eps)
            c2, ll = fb.forward_loglike()
            assert torch.equal(fb.predicted, ref.predicted[:n]) and torch.equal(c2, rc[:n]) and torch.equal(ll, rl[:n]), (eps, w)
            assert torch.equal(fb.forward(), ref.predicted[:n])
    with pytest.raises(AssertionError):
        FdemBatch(s, nl[:4], sig[:4], thk[:4], h[:4], waves=17)


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_jacobian_fixtures_of_imported_reference(golden_npz, name):
    """FdemDataPoint.sensitivity -> nbFdem1dsen fixtures (reference formula), ragged batch, all tensor ids."""
    from geobipy_amd import FdemBatch
    g, s = golden_npz, product_system(name)
    Lmax = 32
    nl, sig, thk, h, Jref = [], [], [], [], []
    for L in [1, 2, 3, 5, 8, 30]:
        k = f"{name}_L{L}"
        n = g[k + "/sigma"].shape[0]
        t = g[k + "/thk"].copy()
        t[:, -1] = 0.0
        nl += [L] * n
        sig.append(pad(g[k + "/sigma"], Lmax, 1.0))
        thk.append(pad(t, Lmax, 0.0))
        h.append(g[k + "/height"])
        Jp = np.zeros((n, g[k + "/J"].shape[1], Lmax))
        Jp[:, :, :L] = g[k + "/J"]
        Jref.append(Jp)
    cat = np.concatenate
    b = FdemBatch(s, np.array(nl), cat(sig), cat(thk), cat(h))
    J = b.sensitivity().cpu().numpy()
    assert close(J, cat(Jref), PRED_ATOL, PRED_RTOL)
    assert np.array_equal(b.sensitivity(max_layers=Lmax).cpu().numpy(), J)      # LDS sizing does not change results


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_fm_dlogc_prediction_and_jacobian_from_one_pass(golden_npz, name):
    """gbp_fdem_fm_dlogc (FdemDataPoint.fm_dlogc): the prediction formed inside the Jacobian sweep against the imported
    reference's predictions (same tolerance as the forward kernel), the Jacobian bit-equal to the Jacobian entry's, soundings
    with 0 layers untouched."""
    from geobipy_amd import FdemBatch
    g, s = golden_npz, product_system(name)
    Lmax = 32
    nl, sig, thk, h, pref = [], [], [], [], []
    for L in [1, 2, 3, 5, 8, 30]:
        k = f"{name}_L{L}"
        t = g[k + "/thk"].copy()
        t[:, -1] = 0.0
        nl += [L] * t.shape[0]
        sig.append(pad(g[k + "/sigma"], Lmax, 1.0))
        thk.append(pad(t, Lmax, 0.0))
        h.append(g[k + "/height"])
        pref.append(g[k + "/pred"])
    cat = np.concatenate
    b = FdemBatch(s, np.array(nl), cat(sig), cat(thk), cat(h))
    for exact in (False, True):
        J = b.fm_dlogc(exact=exact)
        pred = b.predicted.clone()
        assert close(pred.cpu().numpy(), cat(pref), PRED_ATOL, PRED_RTOL)
        assert torch.equal(J, b.sensitivity(exact=exact, bucket=False))
        assert close(pred.cpu().numpy(), b.forward().cpu().numpy(), 1e-9, 1e-12)
    skip = np.array(nl)
    skip[::2] = 0
    b2 = FdemBatch(s, np.array(nl), cat(sig), cat(thk), cat(h))
    b2.nlayers.copy_(torch.as_tensor(skip.astype(np.int32)))
    b2.predicted.fill_(-1.0)
    J2, J = b2.fm_dlogc(), b.fm_dlogc()
    assert torch.all(b2.predicted[::2] == -1.0) and torch.equal(b2.predicted[1::2], pred[1::2]) and torch.equal(J2[1::2], J[1::2])


def test_jacobian_random_batch_vs_oracle_and_finite_differences():
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    osys = oracle_system("syn10")
    B, L = 2048, 6
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=77, Lmax=8)
    b = FdemBatch(s, nl, sig, thk, h)
    J = b.sensitivity().cpu().numpy()
    assert J.shape == (B, 20, 8) and np.all(J[:, :, L:] == 0.0) and np.isfinite(J).all()
    for i in range(0, B, 97):
        Jo = fo.sensitivity(osys, sig[i, :L], thk[i, :L], h[i])
        assert close(J[i, :, :L], np.vstack([Jo.real, Jo.imag]), PRED_ATOL, PRED_RTOL)
    # exact mode = true derivative of the GPU forward (central differences in ln sigma)
    Je = b.sensitivity(exact=True).cpu().numpy()
    eps = 1e-4
    for m in range(L):
        sp, sm = sig.copy(), sig.copy()
        sp[:, m] *= np.exp(eps)
        sm[:, m] *= np.exp(-eps)
        fd = (FdemBatch(s, nl, sp, thk, h).forward().cpu().numpy() - FdemBatch(s, nl, sm, thk, h).forward().cpu().numpy()) / (2 * eps)
        assert np.all(np.abs(Je[:, :, m] - fd) <= 2e-4 + 1e-6 * np.abs(fd))
    # the half-space column agrees between the two modes; the others do not (reference formula)
    assert close(J[:, :, L - 1], Je[:, :, L - 1], PRED_ATOL, 1e-8)


def test_jacobian_of_models_deeper_than_one_evaluation():
    """Launches of more than 16 layers sum four row groups (32 layers) per evaluation (k_fdem_sens<*, 4>); a model of 33 or more layers
    takes a second evaluation for its remaining rows: rows 0 ... 31 and 32 ... against the oracle, both Jacobian expressions, and the
    30-layer case (one evaluation) in the same launch."""
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    osys = oracle_system("syn10")
    Lmax = 44
    nl, sig, thk, h = synthetic.draw_models(48, 40, seed=91, Lmax=Lmax)
    thk[:, :40] = np.where(thk[:, :40] > 0, 1.0 + 0.1 * thk[:, :40], 0.0)           # thin layers: the deep rows stay above rounding level
    nl[::3] = 30; nl[1::3] = 40; nl[2::3] = 44
    sig[2::3, 40:44] = sig[2::3, 36:40][:, ::-1]
    thk[2::3, 39:43] = 2.5
    b = FdemBatch(s, nl, sig, thk, h)
    J = b.sensitivity().cpu().numpy()
    assert J.shape == (48, 20, Lmax) and np.isfinite(J).all()
    for i in range(48):
        L = int(nl[i])
        Jo = fo.sensitivity(osys, sig[i, :L], thk[i, :L], h[i])
        assert close(J[i, :, :L], np.vstack([Jo.real, Jo.imag]), PRED_ATOL, PRED_RTOL), i
        assert np.all(J[i, :, L:] == 0.0)
    assert np.abs(J[1::3, :, 32:40]).max() > 0.0                                      # (the second evaluation's rows are not trivially zero)
    Je = b.sensitivity(exact=True).cpu().numpy()
    eps = 1e-4
    for m in (0, 31, 32, 39):
        sp, sm = sig.copy(), sig.copy()
        sp[:, m] *= np.exp(eps)
        sm[:, m] *= np.exp(-eps)
        fd = (FdemBatch(s, nl, sp, thk, h).forward().cpu().numpy() - FdemBatch(s, nl, sm, thk, h).forward().cpu().numpy()) / (2 * eps)
        live = nl > m
        assert np.all(np.abs(Je[live][:, :, m] - fd[live]) <= 2e-4 + 1e-6 * np.abs(fd[live]))


def test_datapoint_sensitivity_and_fm_dlogc():
    from geobipy_amd import FdemDataPoint, Model, RectilinearMesh1D
    from oracle import fdem_oracle as fo
    mod = Model(mesh=RectilinearMesh1D(edges=np.r_[0.0, 5.0, 7.5, np.inf]), values=np.r_[1e-2, 1e-1, 0.03333333])
    dp = FdemDataPoint(z=30.0, system=os.path.join(GOLDEN, "resolve.stm"))
    dp.fm_dlogc(mod)
    Jo = fo.sensitivity(oracle_system("resolve"), mod.values, [5.0, 2.5, np.inf], 30.0)
    assert dp.sensitivity_matrix.shape == (12, 3)
    assert close(dp.sensitivity_matrix, np.vstack([Jo.real, Jo.imag]), PRED_ATOL, PRED_RTOL)


def test_abscissa_window_mode():
    """FdemBatch's default: a sounding is evaluated with the filter abscissae whose terms can exceed 1e-10 ppm in total at its
    OWN altitude (1 m bins, |rTE| <= 1) -- same results as the full 120-point sums (hankel_eps_ppm=0) to within the budget,
    about half the abscissa points, and bit-identical whatever batch the sounding is evaluated in."""
    from geobipy_amd import FdemBatch, synthetic
    s = synthetic.syn10_system()
    B, L = 8192, 8
    nl, sig, thk, h = synthetic.draw_models(B, L, seed=21)
    h[:5] = [0.3, 3.0, 12.0, 300.0, 2000.0]                     # below / above the usual survey altitudes, beyond the bin range
    exact = FdemBatch(s, nl, sig, thk, h, hankel_eps_ppm=0.0)
    p0 = exact.forward().clone()
    assert exact._h.npoints == 1200 and exact._h.bins is None
    for eps in [None, 1e-12]:
        win = FdemBatch(s, nl, sig, thk, h, hankel_eps_ppm=eps)
        budget = 1e-10 if eps is None else eps
        assert win._h.bins is not None and win._h.bin_points(30.0) < 0.75 * 1200 and win._h.bin_points(30.0) >= 64 * 10
        assert win._h.bin_points(45.0) <= win._h.bin_points(25.0) <= win._h.bin_points(1.0) <= 1200
        p1 = win.forward()
        assert float((p1 - p0).abs().max()) <= budget + 1e-13 * float(p0.abs().max())
    # a sounding's numbers do not depend on its batch (the bins are absolute): any sub-batch, any order, other neighbours
    full = FdemBatch(s, nl, sig, thk, h, waves=2).forward()
    idx = np.random.default_rng(3).permutation(B)[:500]
    sub = FdemBatch(s, nl[idx], sig[idx], thk[idx], h[idx], waves=2).forward()
    assert torch.equal(sub, full[torch.as_tensor(idx, device=full.device)])
    c0, l0 = FdemBatch(s, nl, sig, thk, h, data=np.full((B, 20), 80.0), relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0),
                       hankel_eps_ppm=0.0).forward_loglike()
    c1, l1 = FdemBatch(s, nl, sig, thk, h, data=np.full((B, 20), 80.0), relative_error=np.full(B, 0.05),
                       additive_error=np.full(B, 5.0)).forward_loglike()
    assert torch.allclose(c0, c1, rtol=1e-11, atol=1e-8) and torch.allclose(l0, l1, rtol=1e-11, atol=1e-8)
    # the Jacobian uses the same window: |d rTE / d ln sigma| <= 2/pi bounds the dropped terms of the true derivative by the
    # same sum; the reference's expression (not a derivative) is measured
    for exact_j, bar in ((True, 1e-10), (False, 1e-8)):
        jw = FdemBatch(s, nl[:512], sig[:512], thk[:512], h[:512]).sensitivity(exact=exact_j)
        ja = FdemBatch(s, nl[:512], sig[:512], thk[:512], h[:512], hankel_eps_ppm=0.0).sensitivity(exact=exact_j)
        dj = float((jw - ja).abs().max())
        print("abscissa window, Jacobian (exact=%s): max |dJ| = %.3g ppm per ln(sigma)" % (exact_j, dj))
        assert dj <= bar + 2e-15 * float(ja.abs().max())          # + a few ulp of the largest entries (different summation order)


def test_find_best_halfspace_matches_brute_force():
    """EmDataPoint.find_best_halfspace (100-point log grid, argmin of the misfit) for a whole batch in one launch."""
    from geobipy_amd import FdemBatch, FdemDataPoint, synthetic
    from oracle import fdem_oracle as fo
    s = synthetic.syn10_system()
    osys = oracle_system("syn10")
    B = 16
    nl, sig, thk, h = synthetic.draw_models(B, 1, seed=31)
    clean = FdemBatch(s, nl, sig, thk, h).forward().cpu().numpy()
    obs = synthetic.noisy_observations(clean, seed=32)
    b = FdemBatch(s, nl, sig, thk, h, data=obs, relative_error=np.full(B, 0.05), additive_error=np.full(B, 5.0))
    best, chi2 = b.find_best_halfspace()
    grid = np.logspace(-4, 4, 100)
    for i in range(B):
        phi = [fo.gauss_loglike(fo.predicted_data(osys, [c], [np.inf], h[i]), obs[i], 0.05, 5.0)[1] for c in grid]
        assert np.isclose(best[i].item(), grid[int(np.argmin(phi))], rtol=1e-12)
        assert abs(chi2[i].item() - min(phi)) <= LIKE_ATOL + LIKE_RTOL * min(phi)
    dp = FdemDataPoint(z=h[0], data=obs[0], system=s)
    dp.relative_error, dp.additive_error = 0.05, 5.0
    assert np.isclose(dp.find_best_halfspace().values[0], best[0].item(), rtol=1e-12)


def test_jacobian_bucketing_by_layer_count_is_transparent():
    """Ragged rjMCMC-like population (mostly shallow, a few 30-layer models): bucketed launches = one launch."""
    from geobipy_amd import FdemBatch, synthetic
    rng = np.random.default_rng(8)
    s = product_system("resolve")
    B, Lmax = 2000, 30
    nl = np.where(rng.uniform(size=B) < 0.9, rng.integers(1, 7, size=B), rng.integers(7, Lmax + 1, size=B)).astype(np.int32)
    _, sig, thk, h = synthetic.draw_models(B, Lmax, seed=13)
    b = FdemBatch(s, nl, sig, thk, h)
    assert torch.equal(b.sensitivity(bucket=True), b.sensitivity(bucket=False))


def test_gpu_against_the_analytic_half_space_solution():
    """The kernel on the surface of homogeneous half-spaces (altitude 0, no exponential damping of the filter terms) against
    the closed form (filter accuracy) and against the oracle (parity bar)."""
    from test_oracle_golden import analytic_halfspace_zz_ppm
    from geobipy_amd import FdemBatch, synthetic
    from oracle import fdem_oracle as fo
    ps, osys = synthetic.syn10_system(), oracle_system("syn10")
    sig = np.array([1e-3, 1e-2, 1e-1, 1.0])
    b = FdemBatch(ps, np.ones(4, dtype=np.int32), sig[:, None], np.zeros((4, 1)), np.zeros(4))
    p = b.forward().cpu().numpy()
    F = ps.nFrequencies
    for i, (s_, tol) in enumerate(zip(sig, (1e-2, 5e-3, 2e-3, 5e-4))):
        ana = analytic_halfspace_zz_ppm(ps.frequencies, s_, 7.9)
        assert np.max(np.abs(p[i, :F] + 1j * p[i, F:] - ana) / np.abs(ana)) < tol
        ref = fo.predicted_data(osys, np.array([s_]), np.array([np.inf]), 0.0)
        assert close(p[i], ref, PRED_ATOL, PRED_RTOL)
