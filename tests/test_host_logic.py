"""CPU tier: host logic of the package, the C ABI surface, and the device math headers built for the host.

No compute call through the C ABI happens here (there is no GPU); the numerical scheme of the kernels
(projective recursion, hand-written fp64 sqrt/exp/sincos) is exercised through tests/host_emul, a
test-only g++ build of the very headers the HIP kernels include.
"""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, PRED_ATOL, PRED_RTOL, ROOT, oracle_system

dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int32)


def D(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def I(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ip)


@pytest.fixture(scope="module")
def emul():
    d = os.path.join(ROOT, "tests", "host_emul")
    subprocess.check_call(["make", "-s", "-C", d])
    return ctypes.CDLL(os.path.join(d, "libfdem_host_emul.so"))


def test_capi_library_exports_every_declared_symbol():
    from geobipy_amd import _lib
    from geobipy_amd.build import build_native
    build_native()
    hdr = open(os.path.join(ROOT, "include", "geobipy_amd.h")).read()
    declared = set(re.findall(r"\b(gbp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.load().gbp_version()


def test_missing_library_fails_loudly(monkeypatch):
    from geobipy_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgeobipy_amd.so")
    with pytest.raises(_lib.NativeLibraryError):
        _lib.load()


def test_no_gpu_means_no_silent_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from geobipy_amd import FdemBatch, _lib, synthetic
    with pytest.raises(_lib.NativeLibraryError):
        FdemBatch(synthetic.syn10_system(), *synthetic.draw_models(4, 3))


def test_product_never_imports_oracle():
    """The product path must not import, link or dlopen anything under oracle/ or tests/host_emul."""
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libgbp_oracle|fdem1d_oracle|host_emul|oracle/", re.M)
    for root, _, files in os.walk(os.path.join(ROOT, "geobipy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(root, f)).read()), f


def test_fdem_system_tables_match_oracle_tables():
    from geobipy_amd import FdemSystem
    for name in ["resolve", "mixed", "syn10"]:
        s = FdemSystem.read(os.path.join(GOLDEN, f"{name}.stm"))
        o = oracle_system(name)
        assert np.array_equal(s.tensor_id, o.tid)
        assert np.array_equal(s.loop_separation, o.separation)
        assert np.array_equal(s.lamda0, o.lamda0) and np.array_equal(s.lamda1, o.lamda1)
        a = s.native_args()
        assert np.array_equal(a["scale"], o.scale) and np.array_equal(a["rx_off"], o.rx_off)
    syn = __import__("geobipy_amd").synthetic.syn10_system()
    o = oracle_system("syn10")
    assert np.allclose(syn.frequencies, o.frequencies, rtol=1e-15) and np.all(syn.tensor_id == 9)


def test_datapoint_host_properties():
    from geobipy_amd import FdemDataPoint, FdemSystem
    s = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    d = np.arange(1.0, 13.0)
    d[3] = -2.0
    d[7] = np.nan
    p = FdemDataPoint(z=30.0, data=d, system=s)
    p.relative_error = 0.05
    p.additive_error = 5.0
    assert p.nChannels == 12 and p.n_active_channels == 10
    assert not p.active[3] and not p.active[7]
    ok = ~np.isnan(d)
    assert np.allclose(p.std[ok], np.sqrt((0.05 * d[ok]) ** 2 + 25.0))
    with pytest.raises(AssertionError):
        p.relative_error = -1.0


def test_mesh_and_model():
    from geobipy_amd import Model, RectilinearMesh1D
    m = RectilinearMesh1D(edges=np.r_[0.0, 5.0, 7.5, np.inf])
    assert m.nCells == 3 and np.isinf(m.widths[-1]) and np.allclose(m.widths[:2], [5.0, 2.5])
    m2 = RectilinearMesh1D(edges=np.r_[0.0, -5.0, -7.5, -np.inf])      # heights, as the stitched wedge uses
    assert np.allclose(m2.widths[:2], [5.0, 2.5])
    mod = Model(mesh=m, values=np.r_[1e-2, 1e-1, 0.03])
    assert mod.values.shape == (3,)
    with pytest.raises(AssertionError):
        Model(mesh=m, values=np.ones(4))


def test_partition_rule_matches_reference_example():
    from geobipy_amd.distributed import partition
    starts, sizes = partition(10, 3)            # base/MPI.py:172-201 docstring example
    assert list(starts) == [0, 4, 7] and list(sizes) == [4, 3, 3]
    starts, sizes = partition(65536, 8)
    assert np.all(sizes == 8192) and starts[-1] == 57344


# ---- device math headers, built for the host ---------------------------------------------------------
def test_device_math_accuracy(emul):
    rng = np.random.default_rng(0)
    x = -np.exp(rng.uniform(np.log(1e-8), np.log(745.0), 100000))
    y = np.empty_like(x)
    emul.emul_exp_neg(len(x), D(x)[1], y.ctypes.data_as(dp))
    nrm = x > -700.0                      # below that the result is denormal and carries fewer bits
    assert np.max(np.abs(y - np.exp(x))[nrm] / np.exp(x)[nrm]) < 1e-15
    assert np.max(np.abs(y - np.exp(x))[~nrm]) < 1e-300
    y2 = np.empty(3)
    emul.emul_exp_neg(3, D([-745.0, -800.0, -1e6])[1], y2.ctypes.data_as(dp))
    assert y2[1] == 0.0 and y2[2] == 0.0
    x = rng.uniform(-3000.0, 3000.0, 100000)
    s, c = np.empty_like(x), np.empty_like(x)
    emul.emul_sincos(len(x), D(x)[1], s.ctypes.data_as(dp), c.ctypes.data_as(dp))
    assert np.max(np.abs(s - np.sin(x))) < 5e-16 and np.max(np.abs(c - np.cos(x))) < 5e-16
    a = rng.uniform(-1e-5, 1.0, 100000) * np.exp(rng.uniform(-30, 3, 100000))
    b = np.exp(rng.uniform(np.log(1e-10), np.log(2.0), 100000))
    re, im = np.empty_like(a), np.empty_like(a)
    emul.emul_csqrt(len(a), D(a)[1], D(b)[1], re.ctypes.data_as(dp), im.ctypes.data_as(dp))
    z = np.sqrt(a + 1j * b)
    assert np.max(np.abs((re + 1j * im) - z) / np.abs(z)) < 1e-15
    x = np.exp(rng.uniform(-200, 200, 100000))
    y = np.empty_like(x)
    emul.emul_rcp(len(x), D(x)[1], y.ctypes.data_as(dp))
    assert np.max(np.abs(y * x - 1.0)) < 5e-16


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_kernel_scheme_on_host_vs_reference_fixtures(emul, golden_npz, name):
    """The kernel's formulation (projective recursion, H - H0 accumulated directly) against the fixtures of the
    imported reference, to the parity bar."""
    s = oracle_system(name)
    for L in [1, 2, 3, 5, 8, 30]:
        k = f"{name}_L{L}"
        sig, thk, h = golden_npz[k + "/sigma"], golden_npz[k + "/thk"].copy(), golden_npz[k + "/height"]
        thk[:, -1] = 0.0
        B = sig.shape[0]
        pred = np.empty((B, 2 * s.nF))
        keep = [I(s.tid), D(s.frequencies), D(s.tx_xyz[:, 2]), D(s.rx_xyz[:, 2]), D(s.tx_moment), D(s.scale),
                D(s.rx_off), D(s.separation), D(s.w0), D(s.lamda0), D(s.w1), D(s.lamda1)]
        rc = emul.emul_fdem_forward(s.nF, *[q[1] for q in keep], B, L, I(np.full(B, L))[1], D(sig)[1], D(thk)[1],
                                    D(h)[1], pred.ctypes.data_as(dp))
        assert rc == 0
        ref = golden_npz[k + "/pred"]
        assert np.all(np.abs(pred - ref) <= PRED_ATOL + PRED_RTOL * np.abs(ref)), k


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_jacobian_scheme_on_host_vs_reference_fixtures(emul, golden_npz, name):
    """sens_point (suffix-propagated chain rule) reproduces the reference's nbFdem1dsen fixtures (exact = 0),
    and in exact mode it is the true derivative of the forward solve (central differences of the oracle)."""
    from oracle import fdem_oracle as fo
    s = oracle_system(name)
    keep = [I(s.tid), D(s.frequencies), D(s.tx_xyz[:, 2]), D(s.rx_xyz[:, 2]), D(s.tx_moment), D(s.scale),
            D(s.rx_off), D(s.separation), D(s.w0), D(s.lamda0), D(s.w1), D(s.lamda1)]
    for L in [1, 2, 3, 5, 8, 30]:
        k = f"{name}_L{L}"
        sig, thk, h = golden_npz[k + "/sigma"], golden_npz[k + "/thk"].copy(), golden_npz[k + "/height"]
        thk[:, -1] = 0.0
        B = sig.shape[0]
        J = np.empty((B, 2 * s.nF, L))
        rc = emul.emul_fdem_sens(s.nF, *[q[1] for q in keep], B, L, I(np.full(B, L))[1], D(sig)[1], D(thk)[1],
                                 D(h)[1], 0, J.ctypes.data_as(dp))
        assert rc == 0
        ref = golden_npz[k + "/J"]
        assert np.all(np.abs(J - ref) <= PRED_ATOL + PRED_RTOL * np.abs(ref)), k
    # exact mode vs central differences (L = 5, first two soundings)
    k = f"{name}_L5"
    sig, thk, h = golden_npz[k + "/sigma"][:2], golden_npz[k + "/thk"][:2].copy(), golden_npz[k + "/height"][:2]
    thk[:, -1] = 0.0
    J = np.empty((2, 2 * s.nF, 5))
    emul.emul_fdem_sens(s.nF, *[q[1] for q in keep], 2, 5, I(np.full(2, 5))[1], D(sig)[1], D(thk)[1], D(h)[1], 1,
                        J.ctypes.data_as(dp))
    eps = 1e-4      # the oracle forward carries ~5e-9 ppm rounding noise -> 5e-5 in the difference quotient
    for b in range(2):
        for m in range(5):
            sp, sm = sig[b].copy(), sig[b].copy()
            sp[m] *= np.exp(eps)
            sm[m] *= np.exp(-eps)
            fd = (fo.predicted_data(s, sp, thk[b], h[b]) - fo.predicted_data(s, sm, thk[b], h[b])) / (2 * eps)
            # 'mixed' has ~6e-8 ppm of rounding noise in the oracle forward (non-zero hSum, real-exponent kernels)
            atol = 2e-3 if name == "mixed" else 2e-4
            assert np.all(np.abs(J[b, :, m] - fd) <= atol + 1e-6 * np.abs(fd)), (name, b, m)


@pytest.mark.parametrize("name", ["resolve", "syn10", "mixed"])
def test_abscissa_window_bound_on_host(emul, name):
    """Opt-in accuracy-budgeted abscissa window (gbp_fdem_system_create_windowed): the outputs move by less than
    the budget for every sounding at or above the altitude floor, and about half of the abscissae go."""
    s = oracle_system(name)
    rng = np.random.default_rng(9)
    B, L = 64, 6
    sig = np.exp(rng.uniform(np.log(1e-4), np.log(5.0), (B, L)))
    thk = np.exp(rng.uniform(np.log(0.5), np.log(80.0), (B, L)))
    floor = 20.0
    h = np.r_[floor, rng.uniform(floor, 120.0, B - 1)]
    keep = [I(s.tid), D(s.frequencies), D(s.tx_xyz[:, 2]), D(s.rx_xyz[:, 2]), D(s.tx_moment), D(s.scale),
            D(s.rx_off), D(s.separation), D(s.w0), D(s.lamda0), D(s.w1), D(s.lamda1)]

    def run():
        pred = np.empty((B, 2 * s.nF))
        rc = emul.emul_fdem_forward(s.nF, *[q[1] for q in keep], B, L, I(np.full(B, L))[1], D(sig)[1], D(thk)[1],
                                    D(h)[1], pred.ctypes.data_as(dp))
        assert rc == 0
        return pred, emul.emul_last_npoints()

    emul.emul_set_window.argtypes = [ctypes.c_double, ctypes.c_double]
    emul.emul_set_window(0.0, 0.0)
    full, n_full = run()
    try:
        for eps in [1e-12, 1e-9]:
            emul.emul_set_window(eps, floor)
            win, n_win = run()
            assert np.max(np.abs(win - full)) <= eps + 1e-13 * np.max(np.abs(full))
            assert n_win < 0.75 * n_full
    finally:
        emul.emul_set_window(0.0, 0.0)


def test_sampler_log_and_sincos_headers_built_for_the_host(emul):
    """gbp_math.h log_pos / sincos_quadrant (the sampler's logarithm and Box-Muller angle since round 6), compiled by g++, against
    numpy in long double; the device build of the same source is checked on the hardware by tests/test_gpu_math.py."""
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), 1.0 + rng.uniform(-1e-3, 1e-3, 100000), 1.0 - rng.uniform(0, 1, 100000) ** 8,
                        rng.uniform(0, 1, 100000), [1.0, 2.0, 0.5, 1e-310, 5e-324, 1.7976931348623157e308]])
    out = np.empty_like(x)
    emul.emul_log_pos(ctypes.c_int(x.size), x.ctypes.data_as(dp), out.ctypes.data_as(dp))
    ulp = np.abs(out.astype(np.longdouble) - np.log(x.astype(np.longdouble))) / np.spacing(np.maximum(np.abs(np.log(x)), 1e-300))
    assert float(ulp.max()) < 2.5
    sp, o = np.array([0.0, np.inf, -1.0, np.nan]), np.empty(4)
    emul.emul_log_pos(ctypes.c_int(4), sp.ctypes.data_as(dp), o.ctypes.data_as(dp))
    assert o[0] == -np.inf and o[1] == np.inf and np.isnan(o[2]) and np.isnan(o[3])
    a = np.concatenate([rng.uniform(0, 2 * np.pi, 400000), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi]])
    sn, cs = np.empty_like(a), np.empty_like(a)
    emul.emul_sincos_quadrant(ctypes.c_int(a.size), a.ctypes.data_as(dp), sn.ctypes.data_as(dp), cs.ctypes.data_as(dp))
    al = a.astype(np.longdouble)
    assert float((np.abs(sn - np.sin(al)) / np.spacing(np.maximum(np.abs(np.sin(a)), 1e-300))).max()) < 2.0
    assert float((np.abs(cs - np.cos(al)) / np.spacing(np.maximum(np.abs(np.cos(a)), 1e-300))).max()) < 2.0
