"""The drop-in boundary from a host that is not Python: tests/c_abi/host_demo.cpp (HIP runtime + include/geobipy_amd.h only)
builds the system, evaluates the fused forward + likelihood and runs the whole sampler; its results must equal the Python
host's (DeviceChains) bit for bit -- both are thin drivers of the same C entries."""
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")


def test_cpp_host_compiles_and_links_against_the_header(tmp_path):
    """CPU tier: the demo host builds against include/geobipy_amd.h and the library (no launch)."""
    from geobipy_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("hipcc or the built library is not available")
    exe = str(tmp_path / "host_demo")
    subprocess.check_call([hipcc, "-O1", "-std=c++17", os.path.join(HERE, "c_abi", "host_demo.cpp"), "-o", exe,
                           "-L" + os.path.dirname(_lib.LIB_PATH), "-lgeobipy_amd", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    assert os.path.exists(exe) and subprocess.run([exe]).returncode == 1          # usage error, before any GPU call


@pytest.mark.gpu
def test_cpp_host_runs_the_sampler_through_the_c_abi(tmp_path):
    torch = pytest.importorskip("torch")
    from test_rjmcmc import RESOLVE_OPTIONS
    from geobipy_amd import DeviceChains, FdemSystem, _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "host_demo")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", os.path.join(HERE, "c_abi", "host_demo.cpp"), "-o", exe,
                           "-L" + os.path.dirname(_lib.LIB_PATH), "-lgeobipy_amd", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    s = FdemSystem.read(os.path.join(GOLDEN, "resolve.stm"))
    B, K, n_it, seed = 24, 30, 300, 4242
    rng = np.random.default_rng(3)
    data = np.tile(d["data"], (B, 1)) * rng.uniform(0.8, 1.3, B)[:, None]
    height = rng.uniform(25.0, 40.0, B)
    o = {k: v for k, v in RESOLVE_OPTIONS.items() if k != "n_markov_chains"}
    dc = DeviceChains(s, height, data, seed=seed, exact_jacobian=True, forward_waves=4, **o)      # host_demo.cpp pins 4
    sigma0 = dc.sigma[:, 0].cpu().numpy().copy()
    pred0, chi0, like0 = dc.pred.cpu().numpy().copy(), dc.misfit.cpu().numpy().copy(), dc.like.cpu().numpy().copy()
    dc.run(n_it)
    a, ro = s.native_args(), dc._o
    p = np.array([o["probability_of_birth"], o["probability_of_death"], o["probability_of_perturb"], o["probability_of_no_change"]])
    p = p / p.sum()
    opt = [o["initial_relative_error"], o["initial_additive_error"], ro.min_edge, ro.max_edge, ro.min_width, *p, ro.value_precision,
           ro.gradient_precision, ro.alpha, ro.rel_min[0], ro.rel_max[0], ro.rel_sd[0], ro.add_min[0], ro.add_max[0], ro.add_sd[0], 0.0, 0.0]
    blob = np.concatenate([np.array([s.nFrequencies, B, K, n_it, seed, 0, 0, 0], dtype=np.float64), a["tid"].astype(np.float64),
                           a["frequencies"], a["tx_z"], a["rx_z"], a["tx_moment"], a["scale"], a["rx_off"], a["separation"], a["w0"],
                           a["lamda0"].ravel(), a["w1"], a["lamda1"].ravel(), height, data.ravel(), sigma0, np.array(opt, dtype=np.float64)])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    blob.astype(np.float64).tofile(fin)
    stm = os.path.join(GOLDEN, "SkytemLM.stm")
    out = subprocess.run([exe, fin, fout, stm], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "chains x 300 iterations through the C ABI" in out.stdout and "time domain: 19 windows x 1 component(s), 22 spline nodes" in out.stdout
    r = np.fromfile(fout)
    N = 2 * s.nFrequencies
    off = 0

    def nxt(n):
        nonlocal off
        v = r[off:off + n]
        off += n
        return v
    assert np.array_equal(nxt(B * N).reshape(B, N), pred0) and np.array_equal(nxt(B), chi0) and np.array_equal(nxt(B), like0)
    assert np.array_equal(nxt(B), dc.k.cpu().numpy()) and np.array_equal(nxt(B), dc.n_accepted.cpu().numpy())
    assert np.array_equal(nxt(B * (K + 1)).reshape(B, K + 1), dc.k_hist.cpu().numpy())
    assert np.array_equal(nxt(B * K).reshape(B, K), dc.sigma.cpu().numpy()) and np.array_equal(nxt(B), dc.misfit.cpu().numpy())
    # the time-domain entries from the same host: equal to the Python host's TdemBatch on the same soundings
    from geobipy_amd.tdem import TdemBatch, TdemSystem
    tsig = np.stack([np.full(B, 0.01), np.full(B, 0.1), sigma0], axis=1)
    tthk = np.stack([20.0 + np.arange(B), np.full(B, 40.0), np.zeros(B)], axis=1)
    py = TdemBatch(TdemSystem(stm), np.full(B, 3), tsig, tthk, height, (-13.0, 0.0, 2.0)).forward().cpu().numpy()
    td = nxt(B * 19).reshape(B, 19)
    assert np.abs(td - py).max() <= 1e-10 * np.abs(py).max()
    assert off == r.size and dc.n_accepted.sum() > 1000


@pytest.mark.gpu
def test_two_host_threads_share_one_system_handle(tmp_path):
    """SURVEY 8b "Threading": the compute entries are re-entrant.  tests/c_abi/two_threads.cpp runs two host threads on ONE handle
    with table sets and abscissa windows, each thread with its own stream, batch and per-row set indices (an argument of
    gbp_fdem_forward_rows_ex / gbp_fdem_fm_dlogc_rows_ex; the handle holds no per-call state): 2 x 200 rounds, every result
    bit-equal to the single-threaded one."""
    from geobipy_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "two_threads")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", os.path.join(HERE, "c_abi", "two_threads.cpp"), "-o", exe,
                           "-L" + os.path.dirname(_lib.LIB_PATH), "-lgeobipy_amd", "-lpthread", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatching results" in out.stdout


@pytest.mark.gpu
def test_two_host_threads_share_one_time_domain_handle(tmp_path):
    """The same for the time-domain boundary: gbp_tdem_forward / gbp_tdem_fm_dlogc are re-entrant -- a call leases
    its staging and device scratch from a pool in the handle (by stream), the shared table sets are looked up, grown and handed to the
    launches under the handle's lock.  tests/c_abi/two_threads_tdem.cpp: two threads, own streams, per-row offsets and attitudes,
    one of them meeting new offsets (table growth) while the other's launches are in flight; every result bit-equal to a fresh
    handle's single-threaded one."""
    import numpy as np
    from geobipy_amd import _lib
    from geobipy_amd.filters import W0_J0_120, W1_J1_140
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "two_threads_tdem")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", os.path.join(HERE, "c_abi", "two_threads_tdem.cpp"), "-o", exe,
                           "-L" + os.path.dirname(_lib.LIB_PATH), "-lgeobipy_amd", "-lpthread", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    wfile = str(tmp_path / "weights.bin")
    np.r_[np.asarray(W0_J0_120, dtype=np.float64), np.asarray(W1_J1_140, dtype=np.float64)].tofile(wfile)
    for stm in ("SkytemLM.stm", "tempest.stm"):
        out = subprocess.run([exe, os.path.join(HERE, "golden", stm), wfile], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "0 mismatching results" in out.stdout and "x 60 calls" in out.stdout
