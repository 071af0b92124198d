"""CHECKER for tests/test_hdf5_file.py, run in the BUILD CONTAINER only by an interpreter that has the real h5py AND can import the
reference (/opt/conda/bin/python3.9 there):   python tests/ref_reads_h5.py <file.h5> <index> <out.json>
Opens <file.h5> -- written by geobipy_amd -- with h5py.File and reads sounding <index> with the REFERENCE'S OWN readers
(Histogram.fromHdf, Model.fromHdf, hdfRead.readKeyFromFile -> FdemDataPoint / StatArray / scalars: what Inference2D and
Inference1D.fromHdf call, inversion/Inference2D.py:202-230, 1066-1094, Inference1D.py:1113-1150), then writes what they returned as JSON.
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
from geobipy import Histogram, Model  # noqa: E402
from geobipy.src.base.HDF import hdfRead  # noqa: E402

path, index, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
lst = lambda a: np.asarray(a, dtype=np.float64).ravel().tolist()
res = {"h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version}
with h5py.File(path, "r") as f:
    h = Histogram.fromHdf(f["/model/mesh/nCells/posterior"], index=index)
    res["ncells_posterior_counts"] = lst(h.counts)
    m = Model.fromHdf(f["/model"], index=index)
    res["model"] = dict(nCells=int(np.asarray(m.nCells).item()), values=lst(m.values), edges=[None if not np.isfinite(e) else e for e in lst(m.mesh.edges)])
    hm = Histogram.fromHdf(f["/model/values/posterior"], index=index)
    c = np.asarray(hm.counts)
    res["hitmap"] = dict(shape=list(c.shape), total=int(c.sum()), weighted=float((c * np.arange(c.size).reshape(c.shape)).sum()),
                         x_edges=lst(hm.mesh.x.edges), y_edges=lst(hm.mesh.y.edges))
    eh = Histogram.fromHdf(f["/model/mesh/y/edges/posterior"], index=index)
    res["interface_posterior_counts"] = lst(eh.counts)
    for key in ("iteration", "burned_in", "burned_in_iteration", "multiplier", "halfspace"):
        res[key] = float(np.asarray(hdfRead.readKeyFromFile(f, "", "/", key, index=index)).item())
    dp = hdfRead.readKeyFromFile(f, "", "/", "data", index=index)
    res["datapoint"] = dict(type=type(dp).__name__, data=lst(dp.data), predicted=lst(dp.predictedData), relative_error=lst(dp.relative_error),
                            additive_error=lst(dp.additive_error), z=lst(dp.z), fiducial=lst(dp.fiducial), line_number=lst(dp.line_number),
                            relative_error_posterior_counts=lst(dp.relative_error.posterior.counts),
                            additive_error_posterior_counts=lst(dp.additive_error.posterior.counts))
    ph = np.asarray(hdfRead.readKeyFromFile(f, "", "/", "phids", index=np.s_[index, :]), dtype=np.float64)
    res["phids"] = [None if not np.isfinite(v) else v for v in ph.tolist()]
json.dump(res, open(out, "w"))
