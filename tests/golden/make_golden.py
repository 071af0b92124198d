#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the IMPORTED reference.

Runs only in the build container (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it writes (data only -- inputs and the reference's outputs):
  * resolve.stm, syn10.stm, mixed.stm      system description files (.stm CSV rows)
  * SkytemHM/LM.stm, tempest.stm,          TDEM system files and the reference's TDEM known-answer files
    skytem_/tempest_<type>_clean.csv        (gatdaem1d outputs; the only pin the TDEM path has)
  * resolve_<type>_clean.csv (x6)          copies of the reference's own known-answer files
                                           /root/reference/tests/data_checks/ (test_synthetic_data.py:16-30)
  * fdem_golden.npz                        seeded random soundings per (system, nLayers):
                                           sigma, thk, height, obs, rel, add -> pred[2F], chi2, logL,
                                           J[2F, L] from FdemDataPoint.forward / data_misfit /
                                           likelihood(log=True) / sensitivity of the reference.

The reference is imported with the absent third-party packages stubbed (numba.jit = identity,
so the Numba kernels run interpreted); nothing of the reference is copied into the repo.
"""
import functools
import importlib.abc
import importlib.machinery as im
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SUP = REF + "/documentation_source/source/supplementary/data"


def import_reference():
    missing = {"numba", "h5py", "mpi4py", "cached_property", "pyvista", "pygmt", "numba_kdtree",
               "lmfit", "netCDF4", "progressbar", "randomgen", "gatdaem1d", "empymod"}

    class Stub(types.ModuleType):
        def __getattr__(s, k):
            if k.startswith("__"):
                raise AttributeError(k)
            v = type(k, (), {"__init__": lambda self, *a, **kw: None})
            setattr(s, k, v)
            return v

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(s, name, path, target=None):
            if name.split(".")[0] in missing:
                return im.ModuleSpec(name, s, is_package=True)

        def create_module(s, spec):
            m = Stub(spec.name)
            m.__path__ = []
            return m

        def exec_module(s, m):
            def ident(*a, **k):
                return a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f)
            if m.__name__ == "numba":
                m.jit = m.njit = ident
                m.float64 = float
            if m.__name__ == "numba.pycc":
                m.CC = type("CC", (), {"__init__": lambda s, *a, **k: None,
                                       "export": lambda s, *a, **k: (lambda f: f)})
            if m.__name__ == "cached_property":
                m.cached_property = functools.cached_property

    sys.meta_path.insert(0, Finder())
    gd = im.SourceFileLoader.get_data
    im.SourceFileLoader.get_data = lambda s, p: (
        gd(s, p).replace(b"r_[*xm.shape]", b"r_[tuple(xm.shape)]") if p.endswith("RectilinearMesh2D.py") else gd(s, p))
    import scipy.interpolate.interpnd as _i
    from scipy.interpolate._interpnd import _ndim_coords_from_arrays as _f
    _i._ndim_coords_from_arrays = _f
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    import geobipy
    return geobipy


def write_stm(path, rows):
    with open(path, "w") as f:
        f.write("freq, tor, tmom, tx, ty, tzoff, ror, rmom, rx, ry, rzoff\n")
        for r in rows:
            f.write(", ".join(str(x) for x in r) + "\n")


def main():
    import numpy as np
    from numpy.random import Generator, PCG64DXSM
    gb = import_reference()
    from geobipy import FdemSystem, FdemDataPoint, Model, RectilinearMesh1D

    # --- data files --------------------------------------------------------------------------
    shutil.copyfile(SUP + "/resolve.stm", HERE + "/resolve.stm")
    for f in ["SkytemHM.stm", "SkytemLM.stm", "tempest.stm"]:          # TDEM system files (GA-AEM .stm format)
        shutil.copyfile(SUP + "/" + f, HERE + "/" + f)
    for t in ["glacial", "saline_clay", "resistive_dolomites", "resistive_basement", "coastal_salt_water",
              "ice_over_salt_water"]:                                   # TDEM known answers (test_synthetic_data.py:32-65)
        for fam in ["skytem", "tempest"]:
            shutil.copyfile(REF + f"/tests/data_checks/{fam}_{t}_clean.csv", HERE + f"/{fam}_{t}_clean.csv")
    for t in ["glacial", "saline_clay", "resistive_dolomites", "resistive_basement", "coastal_salt_water",
              "ice_over_salt_water"]:
        shutil.copyfile(REF + f"/tests/data_checks/resolve_{t}_clean.csv", HERE + f"/resolve_{t}_clean.csv")

    # synthetic benchmark system of SURVEY 8(d): 10 zz frequencies, separation 7.9 m
    fr = np.logspace(np.log10(380.0), np.log10(129550.0), 10)
    write_stm(HERE + "/syn10.stm", [[repr(float(f)), "z", 1, 0, 0, 0, "z", 1, 7.9, 0, 0] for f in fr])
    # every tensor id the reference supports (zz=9, xx=1, Tx z/Rx x=3, Tx x/Rx z=7), with
    # vertical loop offsets so that hSum > 0 (the reference overflows for hSum < 0) and a non-unit / negative moment
    write_stm(HERE + "/mixed.stm", [
        [900.0, "z", 1, 0, 0, 0, "z", 1, 7.86, 0, 0],
        [5000.0, "x", -1, 0, 0, 0, "x", 1, 9.06, 0, 0],
        [7230.0, "z", 2, 0, 0, 0.3, "x", 1, 8.1, 0, 0.4],
        [19000.0, "x", 1, 0, 0, 0, "z", 1.5, 7.7, 0, 0.25],
        [56000.0, "z", 1, 0, 0, -0.2, "z", 1, 7.95, 0, 0.1],
        [101000.0, "x", 1, 0, 0, 0.1, "x", -1, 8.8, 0.5, 0.15],
    ])

    out = {}
    seed0 = 20260928
    for si, name in enumerate(["resolve", "syn10", "mixed"]):
        system = FdemSystem.read(HERE + f"/{name}.stm")
        F = system.nFrequencies
        for L in [1, 2, 3, 5, 8, 30]:
            n = 6 if L < 30 else 3
            rng = Generator(PCG64DXSM(seed0 + 100 * si + L))
            sig = np.exp(rng.uniform(np.log(1e-3), np.log(1.0), size=(n, L)))
            thk = np.exp(rng.uniform(np.log(1.0), np.log(50.0), size=(n, L)))
            thk[:, -1] = np.inf
            h = rng.uniform(25.0, 45.0, size=n)
            rel = rng.uniform(0.02, 0.1, size=n)
            add = rng.uniform(1.0, 10.0, size=n)
            pred = np.empty((n, 2 * F))
            obs = np.empty((n, 2 * F))
            chi2 = np.empty(n)
            logl = np.empty(n)
            J = np.empty((n, 2 * F, L))
            for b in range(n):
                edges = np.r_[0.0, np.cumsum(thk[b, :-1]), np.inf]
                mod = Model(mesh=RectilinearMesh1D(edges=edges), values=sig[b].copy())
                dp = FdemDataPoint(x=0.0, y=0.0, z=h[b], elevation=0.0, data=None, std=None, system=system)
                dp.forward(mod)
                clean = np.asarray(dp.predictedData).copy()
                # "observed" data: a perturbed independent model's response would do; use noisy clean
                o = clean * (1.0 + 0.05 * rng.normal(size=2 * F)) + 5.0 * rng.normal(size=2 * F)
                if b % 3 == 2:
                    o[rng.integers(0, 2 * F)] = np.nan       # inactive channel (NaN)
                dp = FdemDataPoint(x=0.0, y=0.0, z=h[b], elevation=0.0, data=o, std=None, system=system)
                dp.relative_error = rel[b]
                dp.additive_error = add[b]
                dp.set_priors(prng=rng)
                dp.forward(mod)
                pred[b] = np.asarray(dp.predictedData)
                obs[b] = o
                chi2[b] = dp.data_misfit()
                logl[b] = dp.likelihood(log=True)
                J[b] = np.asarray(dp.sensitivity(mod))
            key = f"{name}_L{L}"
            for k, v in dict(sigma=sig, thk=thk, height=h, rel=rel, add=add, obs=obs, pred=pred, chi2=chi2,
                             logL=logl, J=J).items():
                out[f"{key}/{k}"] = v
            print(key, "done", flush=True)
    np.savez_compressed(HERE + "/fdem_golden.npz", **out)


if __name__ == "__main__":
    main()
