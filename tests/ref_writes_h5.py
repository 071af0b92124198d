"""CHECKER for tests/test_hdf5_file.py, run in the BUILD CONTAINER only by an interpreter that has the real h5py AND can import the
reference (/opt/conda/bin/python3.9 there):   python tests/ref_writes_h5.py <out.h5> [resolve | skytem | tempest] [iterations to run]
The REFERENCE writes a results container into a real h5py.File: the steps of tests/golden/make_hdf_schema.py (resolve_glacial.csv row 30,
resolve_options, 150 iterations, a three-sounding line, index 1) with ``h5py.File(out, "w")`` where that script hands the reference a
recording stand-in -- Inference1D.createHdf (what Inference2D.createHdf calls per line, Inference2D.py:2001-2015), 150 x accept_reject /
update, Inference1D.writeHdf (Inference1D.py:1002-1090).  The file is what this package's container is compared with, under the real h5py.
The time-domain kinds follow tests/golden/make_tdem_records.py the same way (skytem_glacial.csv / tempest_glacial.csv row 30, 300 / 60
iterations) on tests/golden/fake_gatdaem1d.py -- the stand-in for the absent gatdaem1d whose physics is this repository's oracle: what is
compared is the CONTAINER the reference's TdemDataPoint / Tempest_datapoint createHdf / writeHdf build, not GA-AEM's numbers.
Nothing of geobipy_amd is imported.  Modules the old interpreter lacks are stubbed the way tests/golden/make_golden.py does it;
matplotlib 3.4 gets the ``colormaps`` registry the reference imports."""
import functools
import importlib.abc
import importlib.machinery as im
import json
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
REF = "/root/reference"
MISSING = {"numba", "mpi4py", "cached_property", "pyvista", "pygmt", "numba_kdtree", "lmfit", "netCDF4", "progressbar", "randomgen", "gatdaem1d",
           "empymod", "sklearn", "smm", "cmasher", "cmocean", "cartopy", "pyproj", "rasterio", "shapely", "geopandas"}


class Stub(types.ModuleType):
    def __getattr__(s, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = type(k, (), {"__init__": lambda self, *a, **kw: None})
        setattr(s, k, v)
        return v


class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(s, name, path, target=None):
        if name.split(".")[0] in MISSING:
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
            m.CC = type("CC", (), {"__init__": lambda s, *a, **k: None, "export": lambda s, *a, **k: (lambda f: f)})
        if m.__name__ == "cached_property":
            m.cached_property = functools.cached_property


sys.meta_path.insert(0, Finder())
_gd = im.SourceFileLoader.get_data
im.SourceFileLoader.get_data = lambda s, p: (_gd(s, p).replace(b"r_[*xm.shape]", b"r_[tuple(xm.shape)]") if p.endswith("RectilinearMesh2D.py") else _gd(s, p))
import matplotlib  # noqa: E402
matplotlib.use("Agg")
if not hasattr(matplotlib, "colormaps"):
    import matplotlib.cm as _cm

    class _Registry:
        def __contains__(s, n): return n in _cm._cmap_registry
        def __getitem__(s, n): return _cm.get_cmap(n)
        def __iter__(s): return iter(_cm._cmap_registry)
        def register(s, cmap, name=None, force=False): _cm.register_cmap(name or cmap.name, cmap)
    matplotlib.colormaps = _Registry()
sys.path.insert(0, REF)
import h5py  # noqa: E402
import numpy as np  # noqa: E402
if not hasattr(np, "trapezoid"):                      # (numpy < 2 under the checker; the oracle behind fake_gatdaem1d uses the new name)
    np.trapezoid = np.trapz
kind = sys.argv[2] if len(sys.argv) > 2 else "resolve"
HERE = __import__("os").path.dirname(__import__("os").path.abspath(__file__))
if kind != "resolve":
    sys.path.insert(0, __import__("os").path.dirname(HERE))          # the repository root: oracle/
    sys.path.insert(0, __import__("os").path.join(HERE, "golden"))
    import fake_gatdaem1d  # noqa: E402
    sys.modules["gatdaem1d"] = fake_gatdaem1d
    MISSING.discard("gatdaem1d")
from geobipy import Inference1D, StatArray, get_prng  # noqa: E402
from geobipy.src.inversion import user_parameters as up  # noqa: E402

SUP = REF + "/documentation_source/source/supplementary/data"
OPT = REF + "/documentation_source/source/supplementary/options_files/"
out = sys.argv[1]
if kind == "resolve":
    from geobipy import FdemData
    options = up.user_parameters.read(OPT + "resolve_options", data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    n_it = 150
    ds = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"])
elif kind == "skytem":
    from geobipy import TdemData
    options = up.user_parameters.read(OPT + "skytem_options", data_directory=SUP)
    options["system_filename"] = [SUP + "/SkytemHM.stm", SUP + "/SkytemLM.stm"]
    n_it = 300
    ds = TdemData.read_csv(SUP + "/skytem_glacial.csv", system=options["system_filename"])
else:
    from geobipy import TempestData
    options = up.user_parameters.read(OPT + "tempest_options", data_directory=SUP)
    options["system_filename"] = SUP + "/tempest.stm"
    n_it = 60
    ds = TempestData.read_csv(SUP + "/tempest_glacial.csv", options["system_filename"])
    ds.relative_error = np.tile(np.atleast_1d(options["initial_relative_error"]).astype(float), (ds.nPoints, 1))
    ds.additive_error = np.tile(np.atleast_1d(options["initial_additive_error"]).astype(float), (ds.nPoints, 1))
options.update(n_markov_chains=n_it, save_hdf5=True, interactive_plot=False, update_plot_every=5000)
dp = ds.datapoint(30)
if kind != "resolve":       # (as in make_tdem_records.py: a numpy scalar left in _orientation cannot be written by EmLoop.createHdf; object state only)
    for loop in (dp.loop_pair.transmitter, dp.loop_pair.receiver):
        if not hasattr(loop._orientation, "createHdf"):
            loop._orientation = StatArray(np.atleast_1d(np.asarray(loop._orientation)), "Orientation", dtype=np.int32)
inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
inf.initialize(dp)
fid = np.sort(np.asarray(ds.fiducial)[[29, 30, 31]])
with h5py.File(out, "w") as root:
    inf.createHdf(root, add_axis=fid)
    StatArray(fid).writeHdf(root, "data/fiducial")
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else n_it):      # (optionally fewer iterations than the container is sized for)
        inf.accept_reject()
        inf.update()
    inf.writeHdf(root, index=1)
print("kind", kind, "iteration", int(inf.iteration), "k", int(inf.model.nCells.item()), "h5py", h5py.__version__)
