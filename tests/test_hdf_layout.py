"""the per-line HDF5 results layout.  tests/golden/hdf_schema.json is the tree the REFERENCE's own
Inference2D.createHdf / Inference1D.createHdf / writeHdf produce (recorded through an in-memory stand-in for h5py by
tests/golden/make_hdf_schema.py: every group, dataset, shape, dtype and repr / name / units attribute, plus the values of one
seeded sounding written at index 1 of a three-sounding line).  geobipy_amd.hdf must create the same tree -- entry by entry --
and, with the host sampler (which reproduces the reference's chain decision by decision) on the same sounding and seed, write
the same numbers: counters, best model, traces, and every posterior count."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN
from test_rjmcmc import RESOLVE_OPTIONS, OracleEngine, generator_at


def _run(n_it=150):
    from geobipy_amd import FdemDataPoint, FdemSystem
    from geobipy_amd.inference import Inference1D
    d = np.load(os.path.join(GOLDEN, "mcmc_detail.npz"))
    dp = FdemDataPoint(x=30.0, y=0.0, z=float(d["z"]), elevation=0.0, data=d["data"], system=FdemSystem.read(os.path.join(GOLDEN, "resolve.stm")),
                       lineNumber=0.0, fiducial=30.0)
    dp.engine = OracleEngine("resolve", float(d["z"]))
    o = dict(RESOLVE_OPTIONS, n_markov_chains=n_it, update_plot_every=5000)
    inf = Inference1D(prng=generator_at(d["rng_state"][0]), world=None, save_hdf5=True, reciprocate_parameters=True, **o)
    inf.initialize(dp)
    return inf


def test_layout_and_values_match_the_reference_container():
    from geobipy_amd import hdf
    schema = json.load(open(os.path.join(GOLDEN, "hdf_schema.json")))
    ref, meta = schema["tree"], schema["meta"]
    inf = _run(meta["iterations"])
    root = hdf.NpzGroup("/")
    hdf.create_inference1d(root, inf, add_axis=meta["fiducials"])
    for _ in range(meta["iterations"]):
        inf.accept_reject()
        inf.update()
    inf.writeHdf(root)                                          # index found from the fiducial, like the reference
    ours, arrays = root.walk(), root.arrays()
    assert sorted(ours) == sorted(ref), (sorted(set(ref) - set(ours)), sorted(set(ours) - set(ref)))
    for path, r in ref.items():
        o = ours[path]
        assert o["kind"] == r["kind"], path
        assert {k: str(v) for k, v in o.get("attrs", {}).items()} == r.get("attrs", {}), path
        if r["kind"] != "dataset":
            continue
        assert o["shape"] == r["shape"] and o["dtype"] == r["dtype"], (path, o, r)
        if path in ("/invtime", "/savetime"):                   # wall-clock fields: not written
            continue
        a = arrays[path].astype(np.float64)
        if "values" in r:
            want = np.array([np.nan if v is None else v for v in r["values"]], dtype=np.float64).reshape(a.shape)
            assert np.array_equal(np.isfinite(a), np.isfinite(want)), path
            m = np.isfinite(want)
            exact = arrays[path].dtype.kind in "iub"
            assert np.array_equal(a[m], want[m]) if exact else np.allclose(a[m], want[m], rtol=1e-7, atol=1e-12), (path, a[m][:5], want[m][:5])
        else:
            assert int(np.isfinite(a).sum()) == r["n_finite"], path
            assert np.isclose(np.nansum(a[np.isfinite(a)]), r["nansum"], rtol=1e-9), (path, np.nansum(a[np.isfinite(a)]), r["nansum"])
            if arrays[path].dtype.kind in "iub" and r.get("sha1_of_index_1"):
                assert hashlib.sha1(np.ascontiguousarray(arrays[path][1]).tobytes()).hexdigest() == r["sha1_of_index_1"], path
    assert meta["iteration"] == inf.iteration


def test_infer_writes_through_the_handle_and_the_fallback_file_round_trips(tmp_path):
    """failed = Inference1D(...).infer(hdf_file_handle=group): the harness's call (Inference3D.py:617-620) writes the sounding's
    row when the chain stops; the .npz fallback keeps the HDF paths as entry names."""
    from geobipy_amd import hdf
    inf = _run(60)
    root = hdf.open_results(str(tmp_path / "0.h5"))
    assert isinstance(root, hdf.NpzGroup)                       # h5py is not installed in this image
    hdf.create_inference1d(root, inf, add_axis=[29.0, 30.0, 31.0])
    failed = inf.infer(hdf_file_handle=root)
    assert failed is True and inf.iteration == 60               # never below chi^2 = 12 in 60 iterations: the reference returns True
    assert root["iteration"][1] == 60 and root["iteration"][0] == 0
    assert int(root["model/mesh/nCells/posterior/values/data"][1].sum()) == 60
    root.save(str(tmp_path / "0.npz"))
    z = np.load(str(tmp_path / "0.npz"))
    assert "/model/values/posterior/values/data" in z.files and z["/model/values/posterior/values/data"].shape == (3, 250, 440)
    attrs = json.load(open(str(tmp_path / "0.npz") + ".attrs.json"))
    assert attrs["/model/values"]["repr"] == "StatArray" and attrs["/data"]["repr"] == "FdemData"
