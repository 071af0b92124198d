"""GPU tier: accuracy of the hand-written fp64 kernels on the real hardware (v_rsq_f64 / v_rcp_f64 seeds,
LDS lookup tables, SGPR coefficients), against numpy in fp64/longdouble."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def run(op, x, y=None, two=False):
    from geobipy_amd import _lib
    lib = _lib.load()
    xd = torch.as_tensor(x, dtype=torch.float64, device="cuda")
    yd = None if y is None else torch.as_tensor(y, dtype=torch.float64, device="cuda")
    o0 = torch.empty_like(xd)
    o1 = torch.empty_like(xd) if two else None
    _lib.check(lib.gbp_debug_math(op, xd.numel(), xd.data_ptr(), None if yd is None else yd.data_ptr(),
                                  o0.data_ptr(), None if o1 is None else o1.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return (o0.cpu().numpy(), o1.cpu().numpy()) if two else o0.cpu().numpy()


def test_device_math_accuracy_on_hardware():
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1)
    n = 1 << 20
    x = -np.exp(rng.uniform(np.log(1e-9), np.log(700.0), n))
    e = run(0, x)
    assert np.max(np.abs(e - np.exp(x)) / np.exp(x)) < 1e-15
    assert np.all(run(0, np.array([-746.0, -800.0, -1e6, -1e300])) == 0.0)
    x = rng.uniform(-3000.0, 3000.0, n)
    s, c = run(1, x, two=True)
    assert np.max(np.abs(s - np.sin(x))) < 5e-16 and np.max(np.abs(c - np.cos(x))) < 5e-16
    a = rng.uniform(-1e-5, 1.0, n) * np.exp(rng.uniform(-30, 3, n))
    b = np.exp(rng.uniform(np.log(1e-12), np.log(2.0), n))
    re, im = run(2, a, b, two=True)
    z = np.sqrt(a.astype(np.longdouble) + 1j * b.astype(np.longdouble)) if False else np.sqrt(a + 1j * b)
    assert np.max(np.abs((re + 1j * im) - z) / np.abs(z)) < 1.5e-15
    x = np.exp(rng.uniform(-200, 200, n))
    assert np.max(np.abs(run(3, x) * x - 1.0)) < 5e-16
    g, h = run(6, x, two=True)
    assert np.max(np.abs(g - np.sqrt(x)) / np.sqrt(x)) < 3e-16
    assert np.max(np.abs(h * np.sqrt(x) - 1.0)) < 1e-15


def test_hardware_seed_accuracy_is_what_the_iterations_assume():
    """sqrt_rsqrt / rcp need seeds good to ~2^-20 or better; report what the hardware gives."""
    rng = np.random.default_rng(2)
    x = np.exp(rng.uniform(-100, 100, 1 << 18))
    rs = run(4, x)
    rc = run(5, x)
    e_rsq = np.max(np.abs(rs * np.sqrt(x) - 1.0))
    e_rcp = np.max(np.abs(rc * x - 1.0))
    print("v_rsq_f64 seed max rel err %.3e, v_rcp_f64 seed max rel err %.3e" % (e_rsq, e_rcp))
    assert e_rsq < 2.0 ** -20 and e_rcp < 2.0 ** -20


def test_sampler_log_and_sincos_on_hardware():
    """gbp_math.h log_pos / sincos_quadrant (the sampler's per-chain stages since round 6) on the device, against numpy in long double:
    <= 2.5 ulp for the logarithm over the whole positive range (denormals, the neighbourhood of 1, the special values), <= 2 ulp for sin and
    cos of the Box-Muller angle in [0, 2 pi]."""
    rng = np.random.default_rng(3)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 1 << 19)), 1.0 + rng.uniform(-1e-3, 1e-3, 1 << 18), 1.0 - rng.uniform(0, 1, 1 << 18) ** 8,
                        rng.uniform(0, 1, 1 << 18), [1.0, 2.0, 0.5, 1e-310, 5e-324, 1.7976931348623157e308, 0.7071067811865475, 0.7071067811865476]])
    got = run(7, x)
    ref = np.log(x.astype(np.longdouble))
    ulp = np.abs(got.astype(np.longdouble) - ref) / np.spacing(np.maximum(np.abs(np.log(x)), 1e-300))
    assert float(ulp.max()) < 2.5, (float(ulp.max()), x[int(np.argmax(ulp))])
    sp = run(7, np.array([0.0, np.inf, -1.0, np.nan, 1.0]))
    assert sp[0] == -np.inf and sp[1] == np.inf and np.isnan(sp[2]) and np.isnan(sp[3]) and sp[4] == 0.0
    a = np.concatenate([rng.uniform(0, 2 * np.pi, 1 << 19), 2 * np.pi * rng.integers(0, 2 ** 53, 1 << 18) / 2.0 ** 53,
                        [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 2 * np.pi, np.pi / 4]])
    s, c = run(8, a, two=True)
    al = a.astype(np.longdouble)
    us = np.abs(s - np.sin(al)) / np.spacing(np.maximum(np.abs(np.sin(a)), 1e-300))
    uc = np.abs(c - np.cos(al)) / np.spacing(np.maximum(np.abs(np.cos(a)), 1e-300))
    assert float(us.max()) < 2.0 and float(uc.max()) < 2.0, (float(us.max()), float(uc.max()))
