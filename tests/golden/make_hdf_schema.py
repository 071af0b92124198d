#!/usr/bin/env python
"""The reference's per-line HDF5 results layout (build container only) -> tests/golden/hdf_schema.json.

h5py is not installed here, so the reference's own ``Inference2D.createHdf`` / ``Inference1D.createHdf`` / ``writeHdf``
(inversion/Inference2D.py:2001-2015, inversion/Inference1D.py:1002-1090) are run against a tiny in-memory stand-in for an
h5py group that RECORDS what they ask for: every group, every dataset with its shape / dtype / fill value, every attribute.
One seeded sounding (resolve_glacial.csv row 30, the fixture sounding of mcmc_detail.npz) is inverted for 150 iterations with
the reference's sampler and written into a 3-sounding line container at index 1; the schema holds the tree and, for the
datasets that are small, the values written (model, posterior counts, iteration counters ...), which the host sampler of
geobipy_amd reproduces decision by decision.  Only names / shapes / dtypes / values are stored (data), none of the reference's code.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402

import numpy as np   # noqa: E402


class Attrs(dict):
    pass


class Dataset:
    def __init__(self, name, shape=None, dtype=None, data=None, fillvalue=None, **kw):
        self.name = name
        if data is not None:
            arr = np.array(data)
            self.arr = arr.astype(dtype) if dtype is not None else arr
            if shape is not None:                     # h5py: data given with a shape is reshaped to it (RectilinearMesh1D 'dimension': shape=(1,))
                self.arr = self.arr.reshape((shape,) if np.isscalar(shape) else tuple(int(s) for s in shape))
        else:
            shape = (shape,) if np.isscalar(shape) else tuple(int(s) for s in shape)
            dt = np.dtype(dtype if dtype is not None else "f8")
            self.arr = np.zeros(shape, dtype=dt)
            if fillvalue is not None:
                if dt.kind == "i" and not np.isfinite(fillvalue):
                    self.arr[...] = np.iinfo(dt).min      # what the library's double -> integer conversion of NaN leaves in a real file (x86-64)
                elif dt.kind in "ub" and not np.isfinite(fillvalue):
                    pass                                  # 0
                else:
                    with np.errstate(invalid="ignore"):
                        try:
                            self.arr[...] = fillvalue
                        except (ValueError, TypeError):
                            pass
        self.fillvalue = fillvalue
        self.attrs = Attrs()

    shape = property(lambda s: s.arr.shape)
    dtype = property(lambda s: s.arr.dtype)
    ndim = property(lambda s: s.arr.ndim)
    size = property(lambda s: s.arr.size)

    def __getitem__(self, k):
        return self.arr[k]

    def __setitem__(self, k, v):
        self.arr[k] = v

    def __len__(self):
        return len(self.arr)

    def __array__(self, dtype=None, copy=None):
        return self.arr if dtype is None else self.arr.astype(dtype)


class Group:
    def __init__(self, name="/"):
        self.name, self.items_, self.attrs = name, {}, Attrs()

    def _split(self, path):
        parts = [p for p in path.split("/") if p]
        g = self
        for p in parts[:-1]:
            g = g.items_[p] if p in g.items_ else g.create_group(p)
        return g, parts[-1]

    def create_group(self, name):
        g, last = self._split(name)
        g.items_[last] = Group(g.name.rstrip("/") + "/" + last)
        return g.items_[last]

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def create_dataset(self, name, shape=None, dtype=None, data=None, **kw):
        g, last = self._split(name)
        g.items_[last] = Dataset(g.name.rstrip("/") + "/" + last, shape=shape, dtype=dtype, data=data, **kw)
        return g.items_[last]

    def __getitem__(self, path):
        g = self
        for p in [q for q in path.split("/") if q]:
            g = g.items_[p]
        return g

    def __setitem__(self, path, value):
        self.create_dataset(path, data=value)

    def __contains__(self, path):
        try:
            self[path]
            return True
        except (KeyError, AttributeError):
            return False

    def get(self, path, default=None):
        return self[path] if path in self else default

    def keys(self):
        return self.items_.keys()

    def items(self):
        return self.items_.items()

    def __iter__(self):
        return iter(self.items_)

    def walk(self, out):
        if self.attrs:
            out[self.name] = {"kind": "group", "attrs": {k: (v.decode() if isinstance(v, bytes) else str(v)) for k, v in self.attrs.items()}}
        else:
            out[self.name] = {"kind": "group"}
        for k in sorted(self.items_):
            v = self.items_[k]
            if isinstance(v, Group):
                v.walk(out)
            else:
                e = {"kind": "dataset", "shape": list(v.shape), "dtype": str(v.dtype)}
                # the fill value as a real HDF5 file holds it (checked against a file the reference wrote through the real h5py,
                # tests/ref_writes_h5.py): h5py hands the library a double; NaN -> most negative signed integer, 0 for unsigned / bool
                fv = v.fillvalue
                if fv is not None and np.ndim(fv) == 0:
                    if v.dtype.kind == "f":
                        e["fill"] = "nan" if np.isnan(fv) else float(fv)
                    elif v.dtype.kind == "i":
                        e["fill"] = int(np.iinfo(v.dtype).min) if not np.isfinite(fv) else int(fv)
                    elif v.dtype.kind in "ub":
                        e["fill"] = 0 if not np.isfinite(fv) else int(fv)
                if v.attrs:
                    e["attrs"] = {a: (b.decode() if isinstance(b, bytes) else str(b)) for a, b in v.attrs.items()}
                if v.arr.size <= 512 and v.arr.dtype.kind in "fiub":
                    a = v.arr.astype(np.float64) if v.arr.dtype.kind != "f" else v.arr
                    e["values"] = [None if not np.isfinite(x) else float(x) for x in np.ravel(a)]
                elif v.arr.dtype.kind in "fiub":
                    a = v.arr.astype(np.float64)
                    e["nansum"] = float(np.nansum(a[np.isfinite(a)]))
                    e["n_finite"] = int(np.isfinite(a).sum())
                    import hashlib
                    e["sha1_of_index_1"] = hashlib.sha1(np.ascontiguousarray(v.arr[1]).tobytes()).hexdigest() if v.arr.ndim >= 2 and v.arr.shape[0] == 3 else None
                out[v.name] = e


def main():
    import_reference()
    import h5py                                        # the stub module of import_reference(): give it the names the reference touches
    h5py.Group, h5py.File, h5py.Dataset = Group, Group, Dataset
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    options = up.user_parameters.read(opt_file, data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    options.update(n_markov_chains=150, save_hdf5=True, interactive_plot=False, update_plot_every=5000)
    variant = os.environ.get("GBP_SCHEMA_VARIANT", "")           # "height": the same run with the height sampled -> hdf_schema_height.json
    if variant == "height":
        options.update(solve_z=True, maximum_z_change=1.0, z_proposal_variance=0.01)
    ds = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"])
    dp = ds.datapoint(30)
    inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
    inf.initialize(dp)
    root = Group("/")
    fid = np.sort(np.asarray(ds.fiducial)[[29, 30, 31]])
    inf.createHdf(root, add_axis=fid)                  # Inference2D.createHdf does exactly this, then writes line number / fiducials
    from geobipy import StatArray
    StatArray(fid).writeHdf(root, "data/fiducial")
    for _ in range(150):
        inf.accept_reject()
        inf.update()
    inf.writeHdf(root, index=1)
    tree = {}
    root.walk(tree)
    meta = {"sounding": "resolve_glacial.csv row 30", "iterations": 150, "index": 1, "n_points": 3, "fiducials": [float(x) for x in fid],
            "seed": str(options["seed"]), "iteration": int(inf.iteration), "k": int(inf.model.nCells.item()),
            "note": "tree recorded from the reference's own createHdf / writeHdf through an in-memory stand-in for h5py"}
    if variant:
        meta["variant"] = variant
    json.dump({"meta": meta, "tree": tree}, open(HERE + ("/hdf_schema_" + variant + ".json" if variant else "/hdf_schema.json"), "w"), indent=0, sort_keys=True)
    print(len(tree), "entries;", sum(1 for v in tree.values() if v["kind"] == "dataset"), "datasets")
    for k in sorted(tree):
        v = tree[k]
        print(k, v.get("shape", ""), v.get("dtype", ""), v.get("attrs", ""))


if __name__ == "__main__":
    main()
