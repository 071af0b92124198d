"""Per-line results containers of the reference: the HDF5 layout ``Inference2D.createHdf``
(inversion/Inference2D.py:2001-2015) creates for a flight line -- one row per sounding in every dataset -- and
``Inference1D.writeHdf`` (inversion/Inference1D.py:1039-1090) fills when a sounding's chain stops, so that the reference's own
post-processing reads what this package writes.

The tree (groups, datasets, shapes, dtypes, the ``repr`` / ``name`` / ``units`` attributes the reference's readers dispatch on)
was recorded from the reference's own ``createHdf`` / ``writeHdf`` run against an in-memory stand-in for h5py
(tests/golden/make_hdf_schema.py -> tests/golden/hdf_schema.json); tests/test_hdf_layout.py holds this writer to it entry by
entry -- names, shapes, dtypes, attributes, and the values of a seeded sounding (counters, best model, posterior counts).

``parent`` is anything with the small h5py.Group surface used here (``create_group``, ``create_dataset(name, shape=, dtype=,
data=, fillvalue=)``, ``attrs``, ``__getitem__``): an ``h5py.File`` / ``Group`` when h5py is installed (it is not in this
image), or the bundled ``NpzGroup`` -- same tree, same names, held in memory and saved either as a REAL HDF5 file through the HDF5 C
library (``geobipy_amd.h5lite``: ctypes on a libhdf5 of the image; the file the reference's ``h5py.File`` opens) or, where no such
library exists, as ``<path> -> array`` entries of a ``.npz`` plus a JSON sidecar of the attributes.  ``container_type()`` says which.
"""
import json
import os

import numpy as np

CONDUCTIVITY_UNITS = "$\\frac{S}{m}$"


# ---------------------------------------------------------------------------------------------------------------
# fallback container with the h5py surface the writer uses
# ---------------------------------------------------------------------------------------------------------------
class _Dataset:
    """Dataset of the fallback container.  A dataset created from a shape is LAZY: nothing is allocated until it is written or
    read (a per-line container pre-allocates per-iteration traces -- ``acceptance_rate`` / ``phids``: 2 n_markov_chains values
    per sounding, 1.8 MB at the reference's default -- that the device writer never fills; h5py allocates those lazily on
    disk, and so does this).  Indexing follows h5py's rules for the cases the writer uses, so that what runs here runs there:
    an index array must be strictly increasing (no duplicates) and at most one index array per selection."""

    def __init__(self, name, shape=None, dtype=None, data=None, fillvalue=None):
        self.name, self.attrs = name, {}
        self._arr = None
        if data is not None:
            a = np.array(data)
            self._arr = a.astype(dtype) if dtype is not None else a
            self._shape, self._dtype = self._arr.shape, self._arr.dtype
        else:
            self._shape = (int(shape),) if np.isscalar(shape) else tuple(int(s_) for s_ in shape)
            self._dtype = np.dtype(dtype if dtype is not None else "f8")
        # The fill value as HDF5 holds it.  The reference creates its integer datasets (counters, posterior counts) with fillvalue = NaN
        # like the floating-point ones (DataArray.createHdf, core/DataArray.py:1011-1100); h5py hands the library a double NaN, whose
        # conversion to a signed integer is the most negative value (x86-64; seen in a file the reference wrote, tests/test_hdf5_file.py):
        # rows nothing was written to read -2147483648 / -9223372036854775808 there, and here.  Unsigned and boolean: 0.
        self._fill = None
        if fillvalue is not None:
            if self._dtype.kind == "f":
                self._fill = fillvalue
            elif np.isfinite(fillvalue):
                self._fill = fillvalue
            elif self._dtype.kind == "i":
                self._fill = int(np.iinfo(self._dtype).min)

    shape = property(lambda s: s._shape)
    dtype = property(lambda s: s._dtype)
    materialised = property(lambda s: s._arr is not None)
    sparse = property(lambda s: s._arr is None and bool(getattr(s, "_rle", None)))

    @property
    def arr(self):
        if self._arr is None:
            self._arr = np.zeros(self._shape, dtype=self._dtype)
            if self._fill is not None:
                self._arr[...] = self._fill
            for rows, ptr, start, val in getattr(self, "_rle", None) or ():     # rows written as runs: densify on first dense access
                self._arr.reshape(self._shape[0], -1)[rows] = self._expand(rows.size, ptr, start, val)
            self._rle = None
        return self._arr

    def _expand(self, n_rows, ptr, start, val):
        """[n_rows, M] dense rows from their runs (row j owns runs ptr[j]:ptr[j + 1]; a run lasts to the next run's start / the row's end)."""
        M = int(np.prod(self._shape[1:]))
        g = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(ptr)) * M + start
        return np.repeat(val, np.diff(np.r_[g, n_rows * M])).reshape(n_rows, M)

    def write_run_rows(self, rows, ptr, start, val):
        """Rows ``rows`` (first axis) := their run-length form over the flattened remaining axes: row j owns runs ptr[j]:ptr[j + 1], run
        r holds ``val[r]`` from flat position ``start[r]`` (the first run of a row starts at 0) to the next run's start.  Kept in that
        form while nothing reads or writes the dataset densely: a conductivity-depth hit map is 440 KB per sounding but a few
        thousand runs (depth is the fast axis and a layer fills a run of depth cells of ONE value bin with the same count) --
        survey._LineWriter; ``NpzGroup.save`` writes the runs, ``load_npz`` puts the dense array back."""
        rows, ptr = np.asarray(rows, dtype=np.int64), np.asarray(ptr, dtype=np.int64)
        part = (rows, ptr - ptr[0], np.asarray(start, dtype=np.int32), np.asarray(val).astype(self._dtype, copy=False))
        if self._arr is not None:
            self._arr.reshape(self._shape[0], -1)[rows] = self._expand(rows.size, *part[1:])
            return
        if getattr(self, "_rle", None) is None:
            self._rle = []
        self._rle.append(part)

    def runs(self):
        """(rows, ptr, start, value) of a dataset held in run-length form, rows ascending (rows never written are absent: fill value)."""
        parts = self._rle
        rows = np.concatenate([p_[0] for p_ in parts]) if parts else np.zeros(0, np.int64)
        cnt = np.concatenate([np.diff(p_[1]) for p_ in parts]) if parts else np.zeros(0, np.int64)
        order = np.argsort(rows, kind="stable")
        if parts and not np.array_equal(order, np.arange(rows.size)):
            first = np.concatenate([p_[1][:-1] + off for p_, off in zip(parts, np.r_[0, np.cumsum([p_[2].size for p_ in parts])[:-1]])])
            take = np.concatenate([np.arange(first[j_], first[j_] + cnt[j_]) for j_ in order])
        else:
            take = slice(None)
        start = np.concatenate([p_[2] for p_ in parts]) if parts else np.zeros(0, np.int32)
        val = np.concatenate([p_[3] for p_ in parts]) if parts else np.zeros(0, self._dtype)
        return rows[order].astype(np.int32), np.r_[0, np.cumsum(cnt[order])].astype(np.int64), start[take], val[take]

    @staticmethod
    def _check(k):
        keys = k if isinstance(k, tuple) else (k,)
        lists = [np.asarray(q) for q in keys if isinstance(q, (list, np.ndarray)) and np.ndim(q) > 0 and np.asarray(q).dtype != bool]
        if len(lists) > 1:                                   # h5py's rules (and its exception type), so that what passes here passes there
            raise TypeError("only one indexing vector or array is currently allowed for fancy indexing (h5py)")
        for q in lists:
            if q.size >= 2 and not np.all(np.diff(q) > 0):
                raise TypeError("indexing elements must be in increasing order (h5py)")

    def __getitem__(self, k):
        self._check(k)
        return self.arr[k]

    def __setitem__(self, k, v):
        self._check(k)
        if self._arr is None and not getattr(self, "_rle", None) and self._covers_all(k):
            # the first write covers the whole dataset (a line's rows arriving in one block): one copy instead of allocate + fill + copy
            a = np.empty(self._shape, dtype=self._dtype)
            a[...] = v
            self._arr = a
            return
        self.arr[k] = v

    def _covers_all(self, k):
        keys = k if isinstance(k, tuple) else (k,)
        if len(keys) > len(self._shape):
            return False
        for q, n_ in zip(keys, self._shape):
            if not (isinstance(q, slice) and q.indices(n_) == (0, n_, 1)):
                return False
        return True


class NpzGroup:
    """In-memory group tree with h5py's create_group / create_dataset / attrs / [] surface; ``save(path)`` writes every
    dataset as the entry ``<hdf path>`` of ``path`` (.npz) and the attributes to ``path + '.attrs.json'``."""

    def __init__(self, name="/", container="npz"):
        self.name, self._items, self.attrs = name, {}, {}
        self.container = container               # what save() writes: "npz" (+ .attrs.json) or "hdf5" (h5lite)

    def _child_name(self, last):
        return self.name.rstrip("/") + "/" + last

    def _descend(self, path, create):
        parts = [p for p in path.split("/") if p]
        g = self
        for p in parts[:-1]:
            if p not in g._items:
                if not create:
                    raise KeyError(path)
                g._items[p] = NpzGroup(g._child_name(p))
            g = g._items[p]
        return g, parts[-1]

    def create_group(self, name):
        g, last = self._descend(name, True)
        g._items[last] = NpzGroup(g._child_name(last))
        return g._items[last]

    def create_dataset(self, name, shape=None, dtype=None, data=None, fillvalue=None, **kw):
        g, last = self._descend(name, True)
        g._items[last] = _Dataset(g._child_name(last), shape, dtype, data, fillvalue)
        return g._items[last]

    def __getitem__(self, path):
        g = self
        for p in [q for q in path.split("/") if q]:
            g = g._items[p]
        return g

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def keys(self):
        return self._items.keys()

    def walk(self, out=None):
        """{path: {"kind", "shape", "dtype", "attrs"}} -- the form of tests/golden/hdf_schema.json."""
        out = {} if out is None else out
        out[self.name] = dict(kind="group", **({"attrs": dict(self.attrs)} if self.attrs else {}))
        for k in sorted(self._items):
            v = self._items[k]
            if isinstance(v, NpzGroup):
                v.walk(out)
            else:
                out[v.name] = dict(kind="dataset", shape=list(v.shape), dtype=str(v.dtype), **({"attrs": dict(v.attrs)} if v.attrs else {}))
                if v._fill is not None:            # as tests/golden/hdf_schema*.json record it: "nan" or the number the file holds
                    out[v.name]["fill"] = "nan" if (v.dtype.kind == "f" and np.isnan(v._fill)) else (int(v._fill) if v.dtype.kind in "iub" else float(v._fill))
        return out

    def arrays(self, out=None, materialised_only=False):
        out = {} if out is None else out
        for v in self._items.values():
            if isinstance(v, NpzGroup):
                v.arrays(out, materialised_only)
            elif v.materialised or not materialised_only:          # (a dataset held in sparse form is densified by the access)
                out[v.name] = v.arr
        return out

    def sparse_datasets(self, out=None):
        """{path: dataset} of the datasets held in sparse form (written with ``write_sparse_rows`` only)."""
        out = {} if out is None else out
        for v in self._items.values():
            if isinstance(v, NpzGroup):
                v.sparse_datasets(out)
            elif v.sparse:
                out[v.name] = v
        return out

    def unwritten(self, out=None):
        """{path: (shape, dtype, fill)} of the datasets nothing was ever written to (still lazy)."""
        out = {} if out is None else out
        for v in self._items.values():
            if isinstance(v, NpzGroup):
                v.unwritten(out)
            elif not v.materialised and not v.sparse:
                out[v.name] = dict(shape=list(v.shape), dtype=str(v.dtype), fill=None if v._fill is None else ((int(v._fill) if v.dtype.kind in "iu" else float(v._fill)) if np.isfinite(v._fill) else "nan"))
        return out

    def save(self, path):
        """container "hdf5": the HDF5 file ``path`` (h5lite.write_tree: groups, datasets, attributes as h5py would have written them; datasets
        nothing was written to keep their fill value and take no space; datasets held as runs become chunked + deflated).
        container "npz": ``path`` (.npz): every dataset that was written; ``path + '.attrs.json'``: the attributes and, under the key
        "__unwritten__", shape / dtype / fill value of the datasets that never were (they are all fill value; ``load_npz``
        puts them back) -- so a container costs what was written to it, not what it pre-allocates."""
        if self.container == "hdf5":
            from . import h5lite
            return h5lite.write_tree(path, self)
        lazy = self.unwritten()
        sparse = self.sparse_datasets()          # (before arrays(): a dense access would densify them)
        members = {}
        for name, ds_ in sparse.items():         # "<path>#row" / "#ptr" / "#start" / "#value": the rows' runs; shape in the side file
            members[name + "#row"], members[name + "#ptr"], members[name + "#start"], members[name + "#value"] = ds_.runs()
        dense = self.arrays(materialised_only=True)
        # per-iteration traces ([soundings, slots] float64, NaN beyond the iterations a chain ran): "<path>#len" / "#head" -- the rows'
        # finite prefixes end to end -- when every row is a prefix followed by NaN only; measured misfits do not deflate, NaN padding is free
        tails = {}
        for name in [k for k, a in dense.items() if a.dtype.kind == "f" and a.ndim == 2 and a.nbytes >= (1 << 20)]:
            a = dense[name]
            nan = np.isnan(a)
            ln = a.shape[1] - nan.sum(axis=1)
            if ln.sum() <= 0.75 * a.size and np.array_equal(nan, np.arange(a.shape[1])[None, :] >= ln[:, None]):
                members[name + "#len"], members[name + "#head"] = ln.astype(np.int32), a[~nan]
                tails[name] = dict(shape=list(a.shape), dtype=str(a.dtype))
                del dense[name]
        save_npz(path, dict(dense, **members))
        attrs = {k: v.get("attrs", {}) for k, v in self.walk().items() if v.get("attrs")}
        attrs["__unwritten__"] = lazy
        attrs["__nantail__"] = tails
        attrs["__sparse__"] = {name: dict(shape=list(ds_.shape), dtype=str(ds_.dtype), fill=0 if ds_._fill is None else int(ds_._fill))
                               for name, ds_ in sparse.items()}
        json.dump(attrs, open(str(path) + ".attrs.json", "w"), sort_keys=True)


def _npy_member(name, arr, compresslevel):
    """(member name, method, crc32, raw size, payload) of one array as a zip member holding a .npy file: deflated (raw stream) or STORED.
    Large floating-point members that deflate would shrink by a few % at 150 MB/s (percentile / mean maps, misfit traces: measured values;
    decided on a 64 KB probe from the middle that shrinks by less than 10 %) are stored.  Runs on any thread: zlib releases the lock."""
    import io
    import zlib
    a = np.asanyarray(arr)
    if a.dtype.hasobject:
        raise ValueError("object arrays are not written")
    if a.flags.f_contiguous and not a.flags.c_contiguous:      # written in Fortran order, like numpy.save does: no copy
        flat = a.T
    else:
        a = flat = np.ascontiguousarray(a) if a.ndim else a
    head = io.BytesIO()
    np.lib.format.write_array_header_1_0(head, np.lib.format.header_data_from_array_1_0(a))
    head = head.getvalue()
    body = memoryview(flat.reshape(-1).view(np.uint8)) if a.size else b""
    store = False
    if a.dtype.kind == "f" and a.nbytes >= (1 << 20):
        mid = bytes(body[(a.nbytes // 2) & ~7: ((a.nbytes // 2) & ~7) + 65536])
        store = len(zlib.compress(mid, 1)) > 0.9 * len(mid)
    crc = zlib.crc32(body, zlib.crc32(head))
    raw = len(head) + len(body)
    if store:
        return name + ".npy", 0, crc, raw, [head, body]
    c = zlib.compressobj(compresslevel, zlib.DEFLATED, -15)
    out = [c.compress(head), c.compress(body), c.flush()]
    return name + ".npy", 8, crc, raw, out


def save_npz(path, arrays, compresslevel=1, threads=1):
    """numpy.savez_compressed with a chosen deflate level and the members compressed side by side: the .npz format (a zip of .npy members,
    read back by numpy.load).  Level 1: posterior counts and hit maps are mostly zeros and shrink 100-fold at any level, and at numpy's
    level 6 compressing a flight line's container took ten times as long as inverting the line.  ``threads`` > 1: that many members are
    deflated at a time (a survey's summary file: five 29 MB maps; 0.41 s on one thread for 8 192 soundings), the zip is then put together
    from the finished pieces.  Files beyond the 4 GB / 65 535-member limits of the plain zip records go through zipfile (ZIP64, one thread)."""
    import struct
    file = str(path) if str(path).endswith(".npz") else str(path) + ".npz"
    items = list(arrays.items())
    total = sum(np.asanyarray(v).nbytes + 128 for _, v in items)
    if total >= 0xF0000000 or len(items) >= 0xFFFF:
        return _save_npz_zipfile(file, arrays, compresslevel)
    if threads > 1 and len(items) > 1:
        from concurrent.futures import ThreadPoolExecutor
        order = sorted(range(len(items)), key=lambda q: -np.asanyarray(items[q][1]).nbytes)       # the big ones first
        done = [None] * len(items)
        with ThreadPoolExecutor(max_workers=min(threads, len(items))) as pool:
            for q, r in zip(order, pool.map(lambda q: _npy_member(items[q][0], items[q][1], compresslevel), order)):
                done[q] = r
    else:
        done = [_npy_member(k, v, compresslevel) for k, v in items]
    central, offset = [], 0
    with open(file, "wb") as f:
        for name, method, crc, raw, pieces in done:
            nm = name.encode("utf-8")
            size = sum(len(p_) for p_ in pieces)
            flags = 0x800 if any(ord(ch) > 127 for ch in name) else 0
            f.write(struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, flags, method, 0, 0x21, crc, size, raw, len(nm), 0) + nm)
            for p_ in pieces:
                f.write(p_)
            central.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 20, 20, flags, method, 0, 0x21, crc, size, raw, len(nm), 0, 0, 0, 0, 0x01800000, offset) + nm)
            offset += 30 + len(nm) + size
        cd = b"".join(central)
        f.write(cd)
        f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(done), len(done), len(cd), offset, 0))


def _save_npz_zipfile(file, arrays, compresslevel=1):
    """The same members through zipfile (ZIP64 when needed), one after the other."""
    import zipfile
    with zipfile.ZipFile(file, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=compresslevel, allowZip64=True) as zf:
        for name, arr in arrays.items():
            a = np.asanyarray(arr)
            info = zipfile.ZipInfo(name + ".npy")
            store = False
            if a.dtype.kind == "f" and a.nbytes >= (1 << 20):
                import zlib
                flat = np.ascontiguousarray(a).reshape(-1)
                mid = flat[flat.size // 2: flat.size // 2 + 8192].tobytes()
                store = len(zlib.compress(mid, 1)) > 0.9 * len(mid)
            info.compress_type = zipfile.ZIP_STORED if store else zipfile.ZIP_DEFLATED
            if info.compress_type == zipfile.ZIP_DEFLATED:
                info._compresslevel = compresslevel
            with zf.open(info, "w", force_zip64=True) as member:
                np.lib.format.write_array(member, a, allow_pickle=False)


def load_npz(path):
    """{hdf path: array} of a container written by ``NpzGroup.save``, unwritten datasets expanded to their fill value."""
    out = {}
    side = str(path)[:-4] if str(path).endswith(".npz") else str(path)
    side_file = json.load(open(side + ".attrs.json"))
    sparse = side_file.get("__sparse__", {})
    with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
        for k in z.files:
            if "#" not in k:
                out[k] = z[k]
        for k, m in sparse.items():              # datasets stored as runs (NpzGroup.save): back to dense
            d_ = _Dataset(k, m["shape"], m["dtype"], fillvalue=m.get("fill", 0))
            d_.write_run_rows(z[k + "#row"], z[k + "#ptr"], z[k + "#start"], z[k + "#value"])
            out[k] = d_.arr
        for k, m in side_file.get("__nantail__", {}).items():     # traces stored as their finite prefixes: NaN put back behind them
            a = np.full(m["shape"], np.nan, dtype=np.dtype(m["dtype"]))
            a[np.arange(a.shape[1])[None, :] < z[k + "#len"][:, None]] = z[k + "#head"]
            out[k] = a
    meta = side_file.get("__unwritten__", {})
    for k, m in meta.items():
        a = np.zeros(m["shape"], dtype=np.dtype(m["dtype"]))
        if m["fill"] is not None:
            a[...] = np.nan if m["fill"] == "nan" else m["fill"]
        out[k] = a
    return out


def load_results(path):
    """({hdf path: array}, {hdf path: attributes}) of a flight line's results container of either type: ``<line>.h5`` (read through h5py
    where it is installed, else through the HDF5 C library: h5lite.read_tree -- the reference's own files open the same way) or the
    stand-in ``<line>.results[.npz]``."""
    p_ = str(path)
    if p_.endswith(".h5") or p_.endswith(".hdf5"):
        h5py = _h5py()
        if h5py is None:
            from . import h5lite
            return h5lite.read_tree(p_)
        arrays, attrs = {}, {}
        with h5py.File(p_, "r") as f:
            if len(f.attrs):
                attrs["/"] = dict(f.attrs)

            def visit(name, obj):
                if len(obj.attrs):
                    attrs["/" + name] = dict(obj.attrs)
                if isinstance(obj, h5py.Dataset):
                    arrays["/" + name] = obj[()]
            f.visititems(visit)
        return arrays, attrs
    side = p_[:-4] if p_.endswith(".npz") else p_
    a = json.load(open(side + ".attrs.json"))
    return load_npz(p_), {k: v for k, v in a.items() if not k.startswith("__")}


def _h5py():
    """The h5py module when a usable one is importable (``h5py.File`` a class), else None."""
    try:
        import h5py
        if hasattr(h5py, "File") and isinstance(h5py.File, type):
            return h5py
    except ImportError:
        pass
    return None


def hdf5_writer():
    """What would write an HDF5 file here: "h5py", "libhdf5 <version> through ctypes (<path>)", or None."""
    if _h5py() is not None:
        return "h5py"
    from . import h5lite
    lib = h5lite.load()
    return None if lib is None else "libhdf5 {}.{}.{} through ctypes ({})".format(*lib.version, lib.path)


def container_type(prefer=None):
    """"hdf5" when the results containers are real HDF5 files -- h5py importable, or an HDF5 C library this process can load
    (geobipy_amd.h5lite) --, else "npz": the stand-in with the same dataset paths.  ``prefer`` ("hdf5" | "npz" | "auto" / None), then the
    environment variable GBP_CONTAINER, override the choice; asking for "hdf5" where nothing can write it raises."""
    want = prefer or os.environ.get("GBP_CONTAINER") or "auto"
    if want not in ("auto", "hdf5", "npz"):
        raise ValueError("container must be 'auto', 'hdf5' or 'npz', not {!r}".format(want))
    if want == "npz":
        return "npz"
    if hdf5_writer() is not None:
        return "hdf5"
    if want == "hdf5":
        from . import h5lite
        raise RuntimeError("HDF5 containers were asked for but neither h5py nor a loadable HDF5 library was found: " + str(h5lite.why_not()))
    return "npz"


def results_path(directory, line, container=None):
    """Path ``open_results`` / ``NpzGroup.save`` take for a flight line's container: ``<line>.h5`` for HDF5 -- the reference's file --,
    ``<line>.results`` for the stand-in (saved as ``<line>.results.npz`` + ``<line>.results.attrs.json``: a stand-in is not named .h5)."""
    return os.path.join(str(directory), "{}.{}".format(line, "h5" if container_type(container) == "hdf5" else "results"))


def open_results(path, mode="w", container=None):
    """An ``h5py.File`` when h5py is importable; otherwise an ``NpzGroup`` whose ``.save(path)`` writes the HDF5 file through the HDF5 C
    library (h5lite) or, without one, the .npz stand-in."""
    kind = container_type(container)
    h5py = _h5py()
    if kind == "hdf5" and h5py is not None:
        return h5py.File(path, mode)
    return NpzGroup("/", container=kind)


# ---------------------------------------------------------------------------------------------------------------
# building blocks = the createHdf of the reference's classes (DataArray / StatArray / Histogram / RectilinearMesh*)
# ---------------------------------------------------------------------------------------------------------------
def _attrs(obj, **kw):
    for k, v in kw.items():
        if v is not None:
            obj.attrs[k] = v


def _data_array(parent, name, shape, dtype="f8", fill=np.nan, data=None, rep="DataArray", label=None, units=None, data_fill=None):
    """DataArray.createHdf (core/DataArray.py:1011-1100): group <name> with dataset 'data' and the repr / name / units attributes."""
    g = parent.create_group(name)
    _attrs(g, repr=rep, name=label, units=units)
    if data is not None:
        g.create_dataset("data", data=np.asarray(data, dtype=dtype), **({} if data_fill is None else {"fillvalue": data_fill}))
    else:
        g.create_dataset("data", shape=shape, dtype=dtype, fillvalue=fill)
    return g


def _mesh1d(parent, name, edges, dimension, label="", units="", log=None, relative_to_rows=None, rel_label=None, rel_units=None):
    """RectilinearMesh1D.createHdf: 'dimension', 'edges' (+ 'log' and a per-sounding 'relative_to' for posterior axes that are
    stored relative to a value of the sounding)."""
    g = parent.create_group(name)
    _attrs(g, repr="RectilinearMesh1D")
    g.create_dataset("dimension", data=np.array([dimension], dtype=np.int32))      # (RectilinearMesh1D.py:1641: create_dataset(..., shape=(1,)))
    _data_array(g, "edges", None, data=np.asarray(edges, dtype=np.float64), label=label, units=units)
    if log is not None:
        g.create_dataset("log", data=np.int64(log))
    if relative_to_rows is not None:
        _data_array(g, "relative_to", (relative_to_rows,), label=rel_label, units=rel_units)
    return g


def _index_axis(parent, n):
    """The added first axis of a per-line container: a mesh over the sounding index, edges -0.5 ... n - 0.5."""
    return _mesh1d(parent, "x", np.arange(n + 1) - 0.5, 0)


def _histogram(parent, n, shape, axes, mesh_repr, name="posterior"):
    """Histogram.createHdf: group 'posterior' = mesh (index axis x, then the histogram's own axes) + int32 counts."""
    g = parent.create_group(name)
    _attrs(g, repr="Histogram")
    m = g.create_group("mesh")
    _attrs(m, repr=mesh_repr)
    _index_axis(m, n)
    for name, kw in axes:
        _mesh1d(m, name, **kw)
    _data_array(g, "values", (n,) + tuple(shape), dtype="i4", fill=np.nan, label="Frequency")
    return g


def _stat_array(parent, name, n, shape, label, units, hist_shape, hist_axes, mesh_repr, dtype="f8", fill=np.nan):
    """StatArray.createHdf with one posterior (statistics/StatArray.py:738-800)."""
    g = _data_array(parent, name, (n,) + tuple(shape), dtype=dtype, fill=fill, rep="StatArray", label=label, units=units)
    g.create_dataset("n_posteriors", data=np.int64(1))
    _histogram(g, n, hist_shape, hist_axes, mesh_repr)
    return g


def _stat_array_n(parent, name, n, n_levels, label, units, hist_axes_per_level):
    """StatArray.createHdf of a vector with ONE POSTERIOR PER ENTRY (statistics/StatArray.py:738-800: n_posteriors > 1 ->
    groups posterior0, posterior1, ...): the error levels of a time-domain data point."""
    g = _data_array(parent, name, (n, n_levels) if n_levels > 1 else (n,), rep="StatArray", label=label, units=units)
    g.create_dataset("n_posteriors", data=np.int64(n_levels))
    for i, axes in enumerate(hist_axes_per_level):
        _histogram(g, n, (99,), axes, "RectilinearMesh2D", name="posterior{}".format(i))
    return g


def _loop_rows(parent, name, n, sampled=None):
    """CircularLoops.createHdf with an added axis (system/CircularLoop.py:95-110, EmLoop.py:421-431): one value per sounding.
    ``sampled``: {angle: edges relative to the measured value} for the attitude angles of this loop that are sampled -- those are
    StatArrays with their posterior (EmLoop.set_pitch_posterior ...: tests/golden/hdf_schema_tempest_pitch.json)."""
    g = parent.create_group(name)
    _attrs(g, repr="CircularLoops")
    for key, label, units in (("x", "Easting", "m"), ("y", "Northing", "m"), ("z", "Height", "m"), ("elevation", "Elevation", "m"),
                              ("pitch", "Pitch", "$^{o}$"), ("roll", "Roll", "$^{o}$"), ("yaw", "Yaw", "$^{o}$"), ("moment", "Moment", ""),
                              ("radius", "Radius", "m")):
        if sampled and key in sampled:
            e_ = np.asarray(sampled[key], dtype=np.float64)
            _stat_array(g, key, n, (), label, units, (e_.size - 1,),
                        [("y", dict(edges=e_, dimension=1, label=label, units=units, relative_to_rows=n, rel_label=label, rel_units=units))],
                        "RectilinearMesh2D")
            continue
        _data_array(g, key, (n,), label=label, units=units)
    _data_array(g, "orientation", (n,), dtype="i4", fill=np.nan, label="Orientation")
    return g


def _loop(parent, name, loop):
    g = parent.create_group(name)
    _attrs(g, repr="CircularLoop")
    rows = (("elevation", getattr(loop, "elevation", np.zeros(loop.nPoints)), "Elevation", "m"), ("moment", loop.moment, "Moment", ""),
            ("pitch", loop.pitch, "Pitch", "$^{o}$"), ("radius", loop.radius, "Radius", "m"), ("roll", loop.roll, "Roll", "$^{o}$"),
            ("x", loop.x, "Easting", "m"), ("y", loop.y, "Northing", "m"), ("yaw", loop.yaw, "Yaw", "$^{o}$"), ("z", loop.z, "Height", "m"))
    for key, v, label, units in rows:
        _data_array(g, key, None, data=np.asarray(v, dtype=np.float64), label=label, units=units)
    _data_array(g, "orientation", None, dtype="i4", data=np.asarray(loop._orientation, dtype=np.int32), label="Orientation", units="")


# ---------------------------------------------------------------------------------------------------------------
# Inference1D.createHdf / writeHdf
# ---------------------------------------------------------------------------------------------------------------
def data_kind(dp):
    """'fdem' | 'tdem' | 'tempest': which of the reference's data point classes a container is laid out for."""
    return getattr(dp, "kind", "fdem")


def level_axes(o, kind="fdem"):
    """Posterior axes of the error levels (DataPoint.set_relative_error_posterior / set_additive_error_posterior :651-694;
    Tempest_datapoint.set_additive_error_posterior :533-547): per level (edges [100], relative_to).  Relative and additive
    levels: log10 bins between the prior bounds, stored relative to log10(0.5 (max - min)); the Tempest multiplier: the same bins
    in linear units, relative to 0.5 (max - min)."""
    def one(lo, hi, linear=False):
        lo, hi = np.atleast_1d(np.asarray(lo, dtype=np.float64)), np.atleast_1d(np.asarray(hi, dtype=np.float64))
        out = []
        for a, b in zip(lo, hi):
            e = np.linspace(np.log10(a), np.log10(b), 100)
            if linear:
                to = 0.5 * (b - a)
                out.append((10.0 ** e - to, to))
            else:
                to = np.log10(0.5 * (b - a))
                out.append((e - to, to))
        return out
    return one(o["minimum_relative_error"], o["maximum_relative_error"]), one(o["minimum_additive_error"], o["maximum_additive_error"], kind == "tempest")


def _grids(inf):
    o = inf.options
    p = inf.posteriors
    K = int(o["maximum_number_of_layers"])
    rel, add = level_axes(o, data_kind(inf.datapoint))
    # a sampled height (solve_z): the cells of its uniform prior, stored relative to the measured height (Point.set_z_posterior)
    height = np.linspace(-float(o["maximum_z_change"]), float(o["maximum_z_change"]), 100) if o.get("solve_z") else None
    # sampled attitude angles of a time-domain loop pair: the cells of their uniform priors, relative to the measured angle
    angles = {}
    if data_kind(inf.datapoint) != "fdem":
        from .tdem_geometry import LOOP_PAIR_SCALARS
        for name, stem, nb in LOOP_PAIR_SCALARS:                 # (+ the sampled positions: the pair's offset dx / dy / dz, the transmitter's z)
            if (name[3:] in ("pitch", "roll", "yaw") or name in ("dx", "dy", "dz", "tx_z")) and o.get("solve_" + stem):
                m_ = float(o["maximum_" + stem + "_change"])
                angles[name] = np.linspace(-m_, m_, nb + 1)
    return dict(K=K, rel_edges=rel[0][0], add_edges=add[0][0], rel_to=rel[0][1], add_to=add[0][1], rel_axes=rel, add_axes=add,
                depth_edges=p.depth_edges, value_edges=p.value_edges, value_to=p.relative_to, layer_edges=np.arange(K + 2) - 0.5,
                height_edges=height, angle_edges=angles)


def _create_fdem_data(parent, dp, g_, n, fid):
    """FdemDataPoint.createHdf (data/datapoint/FdemDataPoint.py) under 'data'."""
    N = dp.nChannels
    d = parent.create_group("data")
    _attrs(d, repr="FdemData")
    for key, label, units in (("x", "Easting", "m"), ("y", "Northing", "m"), ("z", "Height", "m"), ("elevation", "Elevation", "m"),
                              ("line_number", "Line number", None)):
        if key == "z" and g_["height_edges"] is not None:      # the height is sampled: a StatArray with its posterior (tests/golden/
            _stat_array(d, key, n, (), label, units, (99,),    #   hdf_schema_height.json: the reference's tree for solve_z = True)
                        [("y", dict(edges=g_["height_edges"], dimension=1, label=label, units=units, relative_to_rows=n, rel_label=label,
                                    rel_units=units))], "RectilinearMesh2D")
        else:
            _data_array(d, key, (n,), label=label, units=units)
    _data_array(d, "fiducial", None, data=fid, label="fiducial", data_fill=np.nan)
    _data_array(d, "data", (n, N), label="Frequency domain data", units="ppm")
    _data_array(d, "std", (n, N), label="Standard deviation", units="ppm")
    _data_array(d, "predicted_data", (n, N), label="Predicted Data", units="ppm")
    for key, label, units, edges, rel_to_units in (("relative_error", "$\\epsilon_{Relative}x10^{2}$", "%", g_["rel_edges"], "%"),
                                                   ("additive_error", "$\\epsilon_{Additive}$", "ppm", g_["add_edges"], "ppm")):
        _stat_array(d, key, n, (), label, units, (99,),
                    [("y", dict(edges=edges, dimension=1, label=label, units=units, log=10, relative_to_rows=n, rel_label=label,
                                rel_units=rel_to_units))], "RectilinearMesh2D")
    s = d.create_group("sys")
    _attrs(s, repr="FdemSystem")
    system = dp.system[0]
    _data_array(s, "freq", None, data=system.frequencies, label="Frequencies", units="Hz")
    _loop(s, "T", system.transmitter)
    _loop(s, "R", system.receiver)


TD_UNITS = {"tdem": "$\\frac{V}{m^{2}}$", "tempest": "fT"}


def _create_tdem_data(parent, dp, g_, n, fid, kind):
    """TdemDataPoint.createHdf (data/datapoint/TdemDataPoint.py:603-625) / Tempest_datapoint.createHdf (Tempest_datapoint.py:566-
    572) under 'data', as recorded in tests/golden/hdf_schema_tdem.json: the DataPoint part (location, data, std, predicted data,
    error levels with one posterior per level), the systems' .stm texts, the component indices, the loop pair (offset +
    transmitter + receiver with attitude), primary / secondary fields and their predictions; Tempest: per-channel additive
    errors and their per-component multiplier."""
    N, units, tempest = dp.nChannels, TD_UNITS[kind], kind == "tempest"
    d = parent.create_group("data")
    _attrs(d, repr="TempestData" if tempest else "TdemData")
    for key, label, u in (("x", "Easting", "m"), ("y", "Northing", "m"), ("z", "Height", "m"), ("elevation", "Elevation", "m"),
                          ("line_number", "Line number", None)):
        _data_array(d, key, (n,), label=label, units=u)
    _data_array(d, "fiducial", None, data=fid, label="fiducial", data_fill=np.nan)
    _data_array(d, "data", (n, N), label="Data" if tempest else "Secondary field", units=units)
    _data_array(d, "std", (n, N), label="Standard deviation", units=units)
    _data_array(d, "predicted_data", (n, N), label="Predicted Data" if tempest else "Predicted secondary field", units=units)
    rel_label = "$\\epsilon_{Relative}$" if tempest else "$\\epsilon_{Relative}x10^{2}$"
    axes = lambda levels, label, u, log: [[("y", dict(edges=e, dimension=1, label=label, units=u, log=log, relative_to_rows=n, rel_label=label,
                                                       rel_units=u))] for e, _ in levels]
    n_rel, n_sys = len(g_["rel_axes"]), len(dp.system)
    _stat_array_n(d, "relative_error", n, n_rel, rel_label, "%", axes(g_["rel_axes"], rel_label, "%", 10))
    if tempest:
        _data_array(d, "additive_error", (n, N), label="$\\epsilon_{additive}$", units=units)
        _stat_array_n(d, "additive_error_multiplier", n, len(g_["add_axes"]), "Multiplier", None, axes(g_["add_axes"], "Multiplier", None, None))
    else:
        _stat_array_n(d, "additive_error", n, len(g_["add_axes"]), "$\\epsilon_{Additive}$", units, axes(g_["add_axes"], "$\\epsilon_{Additive}$", units, 10))
    d.create_dataset("nSystems", data=np.int64(n_sys))
    for i, s_ in enumerate(dp.system):
        g = d.create_group("System{}".format(i))
        with open(s_.filename) as f:
            _attrs(g, repr="TdemSystem", data=f.readlines())
    d.create_dataset("components", data=np.asarray(["xyz".index(c) for c in dp.system[0].components], dtype=np.int32))
    lp = d.create_group("loop_pair")
    _attrs(lp, repr="Loop_pair")
    ang = g_.get("angle_edges") or {}
    for key, label, u in (("x", "Easting", "m"), ("y", "Northing", "m"), ("z", "Height", "m"), ("elevation", "Elevation", "m")):
        if "d" + key in ang:         # a sampled component of the pair's offset (the receiver's x / y / z priors go to the offset, Loop_pair.set_priors
            e_ = np.asarray(ang["d" + key], dtype=np.float64)      # :172-178): a StatArray with its posterior on the prior's cells, like a sampled angle
            _stat_array(lp, key, n, (), label, u, (e_.size - 1,),
                        [("y", dict(edges=e_, dimension=1, label=label, units=u, relative_to_rows=n, rel_label=label, rel_units=u))], "RectilinearMesh2D")
        else:
            _data_array(lp, key, (n,), label=label, units=u)
    _loop_rows(lp, "transmitter", n, {k_[3:]: v_ for k_, v_ in ang.items() if k_.startswith("tx_")})
    _loop_rows(lp, "receiver", n, {k_[3:]: v_ for k_, v_ in ang.items() if k_.startswith("rx_")})
    nc = dp.system[0].n_components
    wide = lambda m: (n, m) if m > 1 else (n,)
    _data_array(d, "primary_field", wide(n_sys * nc), label="Primary field", units=units)
    _data_array(d, "secondary_field", (n, N), label="Secondary field", units=units)
    _data_array(d, "predicted_primary_field", wide(nc), label="Predicted primary field", units=units)
    _data_array(d, "predicted_secondary_field", (n, N), label="Predicted secondary field", units=units)


def create_inference1d(parent, inf, add_axis):
    """Inference1D.createHdf(parent, add_axis=fiducials) as Inference2D.createHdf calls it, plus that method's own two writes
    (line number, sorted fiducials).  ``inf``: an initialised geobipy_amd Inference1D; ``add_axis``: the line's fiducials."""
    fid = np.sort(np.atleast_1d(np.asarray(add_axis, dtype=np.float64)))
    n = fid.size
    dp, g_ = inf.datapoint, _grids(inf)
    N, K = dp.nChannels, g_["K"]
    kind = data_kind(dp)
    if kind == "fdem":
        _create_fdem_data(parent, dp, g_, n, fid)
    else:
        _create_tdem_data(parent, dp, g_, n, fid, kind)

    parent.create_dataset("update_plot_every", data=np.int32(inf.options.get("update_plot_every") or 5000))
    parent.create_dataset("interactive_plot", data=np.bool_(inf.interactive_plot))
    parent.create_dataset("reciprocate_parameter", data=np.bool_(inf.reciprocate_parameter))
    if inf.options.get("parameter_limits") is not None:
        parent.create_dataset("limits", data=np.asarray(inf.options["parameter_limits"], dtype=np.float64))
    parent.create_dataset("n_markov_chains", data=np.int64(inf.n_markov_chains))
    parent.create_dataset("nsystems", data=np.int64(dp.nSystems))
    # (the reference creates all of them with fillvalue = NaN, Inference1D.createHdf :1010-1030: the counters' rows nothing wrote to hold the
    #  most negative int64 in its files -- _Dataset -- and so do these)
    for key, dt, fill in (("iteration", "i8", np.nan), ("burned_in_iteration", "i8", np.nan), ("best_iteration", "i8", np.nan), ("burned_in", "?", 0),
                          ("multiplier", "f8", np.nan), ("invtime", "f8", np.nan), ("savetime", "f8", np.nan)):
        parent.create_dataset(key, shape=(n,), dtype=dt, fillvalue=fill)
    # per-iteration traces (Inference1D.acceptance_v / data_misfit_v :408, 414): 2 n_markov_chains entries per sounding.  A device
    # block may keep every `trace_every`-th entry only (LineSpec.trace_every > 1): the datasets then hold those entries side by
    # side -- ceil(2 n_markov_chains / trace_every) columns -- and carry the stride as their attribute `trace_every`
    every = int(getattr(inf, "trace_every", 1) or 1)
    n_tr = -(-2 * inf.n_markov_chains // every)
    ga = _data_array(parent, "acceptance_rate", (n, n_tr), dtype="u1", fill=None, label="% Acceptance")
    gp = _data_array(parent, "phids", (n, n_tr), label="Data Misfit")
    if every > 1:
        _attrs(ga, trace_every=every)
        _attrs(gp, trace_every=every)
    _data_array(parent, "halfspace", (n,), label="halfspace", units=CONDUCTIVITY_UNITS)

    m = parent.create_group("model")
    _attrs(m, repr="Model")
    mesh = m.create_group("mesh")
    _attrs(mesh, repr="RectilinearMesh2D_stitched")
    _index_axis(mesh, n)
    _stat_array(mesh, "nCells", n, (), "Number of cells", None, (K + 1,),
                [("y", dict(edges=g_["layer_edges"], dimension=0, label="# of Layers", units=""))], "RectilinearMesh2D", dtype="i4", fill=np.nan)
    y = mesh.create_group("y")
    e = _data_array(y, "edges", (n, K + 1), rep="StatArray")
    e.create_dataset("n_posteriors", data=np.int64(1))
    _histogram(e, n, (g_["depth_edges"].size - 1,), [("y", dict(edges=g_["depth_edges"], dimension=0, label="Depth", units="m"))],
               "RectilinearMesh2D")
    _data_array(y, "relative_to", (n,))
    _stat_array(m, "values", n, (K,), "Conductivity", CONDUCTIVITY_UNITS, (g_["value_edges"].size - 1, g_["depth_edges"].size - 1),
                [("y", dict(edges=g_["value_edges"], dimension=1, label="Conductivity", units=CONDUCTIVITY_UNITS, log=10, relative_to_rows=n)),
                 ("z", dict(edges=g_["depth_edges"], dimension=2, label="Depth", units="m"))], "RectilinearMesh3D")
    return parent


def write_inference1d(parent, inf, index=None):
    """Inference1D.writeHdf: this sounding's row of every dataset -- counters, traces, the posteriors of the chain, and the
    highest-posterior data point / model (which the reference writes last, over the current ones)."""
    dp = inf.datapoint
    if index is None:
        index = int(np.searchsorted(np.asarray(parent["data/fiducial/data"][...]), float(np.ravel(dp.fiducial)[0])))
    i, g_ = index, _grids(inf)
    n_mc, K = inf.n_markov_chains, g_["K"]
    parent["iteration"][i] = inf.iteration
    parent["burned_in_iteration"][i] = int(getattr(inf, "burned_in_iteration", 0))
    parent["best_iteration"][i] = inf.best_iteration
    parent["burned_in"][i] = bool(getattr(inf, "burned_in", False))
    parent["multiplier"][i] = inf.multiplier
    # the reference stores the flag of update m at index m (Inference1D.update :726) and the misfit at m - 1 (:716)
    parent["acceptance_rate/data"][i, :] = inf.acceptance_v[:2 * n_mc]
    parent["phids/data"][i, :] = inf.data_misfit_v[:2 * n_mc]
    parent["halfspace/data"][i] = float(inf.halfspace[0])
    best, p = inf.best_state, inf.posteriors
    d = parent["data"]
    for key, v in (("x", dp.x), ("y", dp.y), ("z", dp.z[0]), ("elevation", dp.elevation), ("line_number", np.ravel(dp.lineNumber)[0])):
        d[key + "/data"][i] = float(v)
    data = np.asarray(dp.data, dtype=np.float64)
    if data_kind(dp) != "fdem":
        _write_tdem_point(d, i, dp, data, best, p, g_, inf.error_model)
    else:
        d["data/data"][i, :] = data
        d["std/data"][i, :] = np.sqrt((best.rel * data) ** 2.0 + best.add ** 2.0)
        d["predicted_data/data"][i, :] = best.pred
        d["relative_error/data"][i] = best.rel
        d["additive_error/data"][i] = best.add
        d["relative_error/posterior/values/data"][i, :] = p.relative_error
        if g_["height_edges"] is not None:                     # the best data point's height, the height posterior about the measured one
            d["z/data"][i] = best.z
            d["z/posterior/values/data"][i, :] = p.height
            d["z/posterior/mesh/y/relative_to/data"][i] = inf.z_move.z0
        d["additive_error/posterior/values/data"][i, :] = p.additive_error
        d["relative_error/posterior/mesh/y/relative_to/data"][i] = g_["rel_to"]
        d["additive_error/posterior/mesh/y/relative_to/data"][i] = g_["add_to"]
    m = parent["model"]
    k = best.values.size
    m["mesh/nCells/data"][i] = k
    m["mesh/nCells/posterior/values/data"][i, :] = p.n_cells
    row = np.full(K + 1, np.nan)
    row[:k + 1] = np.r_[0.0, best.edges, np.inf]
    m["mesh/y/edges/data"][i, :] = row
    m["mesh/y/edges/posterior/values/data"][i, :] = p.edges
    vals = np.full(K, np.nan)
    vals[:k] = best.values
    m["values/data"][i, :] = vals
    m["values/posterior/values/data"][i, :, :] = p.values
    m["values/posterior/mesh/y/relative_to/data"][i] = g_["value_to"]


def _sampled_scalar_group(name):
    """Container group of a sampled scalar of the loop pair: tx_pitch -> loop_pair/transmitter/pitch, dz -> loop_pair/z (the offset)."""
    if name in ("dx", "dy", "dz"):
        return "loop_pair/" + name[1]
    return "loop_pair/{}/{}".format("transmitter" if name.startswith("tx_") else "receiver", name[3:])


def _write_loop_pair(d, i, x, y, z, offset, loop_angles, radius):
    """Loop_pair.writeHdf (system/Loop_pair.py:305-315): the pair's offset, the transmitter at the sounding, the receiver at
    transmitter + offset, both loops' pitch / roll / yaw (the file's convention), radius, unit moment, z orientation."""
    lp = d["loop_pair"]
    for key, v in (("x", offset[0]), ("y", offset[1]), ("z", offset[2]), ("elevation", 0.0)):
        lp[key + "/data"][i] = v
    a = np.asarray(loop_angles, dtype=np.float64)
    for name, dx, ang in (("transmitter", (0.0, 0.0, 0.0), a[..., 0:3]), ("receiver", offset, a[..., 3:6])):
        g = lp[name]
        for key, v in (("x", x + dx[0]), ("y", y + dx[1]), ("z", z + dx[2]), ("elevation", 0.0), ("pitch", ang[..., 0]), ("roll", ang[..., 1]),
                       ("yaw", ang[..., 2]), ("moment", 1.0), ("radius", radius)):
            g[key + "/data"][i] = v
        g["orientation/data"][i] = 2


def _write_tdem_point(d, i, dp, data, best, p, g_, error_model):
    """TdemDataPoint.writeHdf (data/datapoint/TdemDataPoint.py:627-645) / Tempest_datapoint.writeHdf for one host-sampled sounding."""
    rel, add = np.atleast_1d(best.rel), np.atleast_1d(best.add)
    tempest = data_kind(dp) == "tempest"
    off = dp.offset
    nc = dp.system[0].n_components
    best_geom = getattr(best, "geom", None) or {}
    att = dp.attitude
    if best_geom:                                             # a sampled geometry: the best data point's loop pair
        from .tdem_geometry import gaaem_tuple, loop_pair_values
        g10 = gaaem_tuple(dict(loop_pair_values(dp.transmitter, dp.receiver), **best_geom))
        off, att = tuple(g10[4:7]), tuple(np.r_[g10[1:4], g10[7:10]])
    pf = np.concatenate([s_.primary_field(*off, attitude=att) for s_ in dp.system])                  # predicted primary field per component
    per = [s_.nwindows for s_ in dp.system for _ in range(s_.n_components)]
    d["data/data"][i, :] = data
    d["std/data"][i, :] = error_model.std(data, rel, add)
    d["predicted_data/data"][i, :] = best.pred
    if tempest:          # total-field channels: Tempest_datapoint.py:106-123 (data / predictedData), :94-104 (multipliers)
        d["secondary_field/data"][i, :] = dp.secondary_field
        d["predicted_secondary_field/data"][i, :] = best.pred - np.repeat(pf, per)
        d["additive_error/data"][i, :] = dp.channel_additive_error
        d["additive_error_multiplier/data"][i] = add
        add_name = "additive_error_multiplier"
    else:
        d["secondary_field/data"][i, :] = data
        d["predicted_secondary_field/data"][i, :] = best.pred
        d["additive_error/data"][i] = add if add.size > 1 else add[0]
        add_name = "additive_error"
    d["relative_error/data"][i] = rel if rel.size > 1 else rel[0]
    for g in range(rel.size):
        d["relative_error/posterior{}/values/data".format(g)][i, :] = np.atleast_2d(p.relative_error)[g]
        d["relative_error/posterior{}/mesh/y/relative_to/data".format(g)][i] = g_["rel_axes"][g][1]
    for g in range(add.size):
        d[add_name + "/posterior{}/values/data".format(g)][i, :] = np.atleast_2d(p.additive_error)[g]
        d[add_name + "/posterior{}/mesh/y/relative_to/data".format(g)][i] = g_["add_axes"][g][1]
    ang = [float(np.atleast_1d(getattr(lp_, k_))[0]) for lp_ in (dp.transmitter, dp.receiver) for k_ in ("pitch", "roll", "yaw")]
    for j_, name in enumerate(("tx_pitch", "tx_roll", "tx_yaw", "rx_pitch", "rx_roll", "rx_yaw")):
        if name in best_geom:                                   # a sampled angle: the best data point's (Inference1D.writeHdf :1076-1088)
            ang[j_] = float(best_geom[name])
    _write_loop_pair(d, i, float(dp.x), float(dp.y), float(dp.z[0]), off, ang, dp.system[0].loopRadius())
    for name in (g_.get("angle_edges") or {}):
        grp = _sampled_scalar_group(name)
        d[grp + "/posterior/values/data"][i, :] = p.geometry[name]
        d[grp + "/posterior/mesh/y/relative_to/data"][i] = p.geometry_edges[name][0] + 0.5 * (p.geometry_edges[name][-1] - p.geometry_edges[name][0])
    d["primary_field/data"][i] = getattr(dp, "primary_field", np.zeros(pf.size)) if pf.size > 1 else 0.0
    d["predicted_primary_field/data"][i] = pf[:nc] if nc > 1 else pf[0]


# ---------------------------------------------------------------------------------------------------------------
# Per-line containers from the device sampler (survey.infer(results_directory=...)): the same tree, one row per sounding,
# filled from the chains' device-resident posteriors instead of a host Inference1D
# ---------------------------------------------------------------------------------------------------------------
class LineSpec:
    """What ``create_inference1d`` reads of an initialised Inference1D, built from a DeviceChains block instead: the
    acquisition system(s), the options, the kind of data point ('fdem' | 'tdem' | 'tempest') and the posterior grids (the device
    accumulators bin on the reference's grids: RectilinearMesh1D.set_posteriors :1438-1455, Model.set_posteriors)."""

    def __init__(self, system, n_channels, options, n_value_bins=250, kind="fdem", trace_every=1):
        from types import SimpleNamespace
        from .inference import OPTION_DEFAULTS
        o = dict(OPTION_DEFAULTS)
        o.update({k: v for k, v in dict(options).items() if v is not None})
        self.options = o
        systems = list(system) if isinstance(system, (list, tuple)) else [system]
        self.datapoint = SimpleNamespace(nChannels=int(n_channels), system=systems, nSystems=len(systems), kind=kind)
        min_width = float(o["minimum_thickness"])
        half = 4.0 * np.log(1.0 + float(o["factor"]))
        self.posteriors = SimpleNamespace(depth_edges=np.arange(0.0, 1.1 * float(o["maximum_depth"]), 0.5 * min_width),
                                          value_edges=np.linspace(-half, half, int(n_value_bins) + 1) / np.log(10.0), relative_to=np.nan)
        self.interactive_plot = False
        self.reciprocate_parameter = bool(o.get("reciprocate_parameters", True))
        self.n_markov_chains = int(o["n_markov_chains"])
        self.trace_every = int(trace_every or 1)       # stride of the per-iteration traces the device block kept (1: the reference's arrays)


# per-sounding fields of a finished block, as survey.infer ships them to the writing rank: (name, columns, kind)
def device_row_fields(N, K, n_depth, n_value, n_err=99, hitmap=True, n_rel=1, n_add=1, time_domain=False, n_primary=0, height=False,
                      angles=(), trace_length=0):
    """``n_rel`` / ``n_add``: error levels per sounding (time-domain data: one relative level per system x component, one additive
    level -- or Tempest multiplier -- per system / component); ``time_domain``: the loop pair's offset and both loops' angles, the
    per-channel standard deviation and the (file, predicted) primary fields travel too."""
    f64 = [("x", 1), ("y", 1), ("z", 1), ("elevation", 1), ("line_number", 1), ("fiducial", 1), ("data", N), ("predicted", N),
           ("relative_error", n_rel), ("additive_error", n_add), ("log_mean_prior", 1), ("best_edges", K), ("best_sigma", K)]
    if time_domain:
        f64 += [("std", N), ("offset", 3), ("loop_angles", 6), ("primary", n_primary), ("predicted_primary", n_primary)]
    i32 = [("status", 1), ("burned_in_iteration", 1), ("iterations", 1), ("best_k", 1), ("best_iteration", 1), ("k_hist", K + 1), ("edge_hist", n_depth),
           ("rel_hist", n_rel * n_err), ("add_hist", n_add * n_err)] + ([("hitmap", n_value * n_depth)] if hitmap else [])
    if height:                                       # a sampled height (solve_z): the best state's, the measured one, the posterior
        f64 += [("best_height", 1), ("height0", 1)]
        i32 += [("height_hist", n_err)]
    for name, nb in angles:                          # sampled attitude angles [(name, cells)]: the same three per angle
        f64 += [("best_" + name, 1), (name + "_centre", 1)]
        i32 += [(name + "_hist", nb)]
    if trace_length:                                 # the (decimated) per-iteration traces: misfit, accept / reject decisions
        f64 += [("trace_misfit", trace_length)]
        i32 += [("trace_accept", trace_length)]
    return f64, i32


def write_device_rows(parent, index, f64, i32, N, K, n_depth, n_value, options, hitmap=True, kind="fdem", n_rel=1, n_add=1, n_primary=0,
                      loop_radius=0.0, channel_additive=None, height=False, angles=(), trace_length=0, hitmap_csr=None):
    """Rows ``index`` (positions along the line's sorted fiducials) of a container made by ``create_inference1d(parent,
    LineSpec(...), fiducials)``, from the two blocks of ``device_row_fields`` (numpy, one row per sounding).  The per-iteration
    ``hitmap_csr`` = (ptr [m + 1], start, value): the rows' hit maps in run-length form (flat position value_bin * n_depth + depth cell;
    _Dataset.write_run_rows) instead of dense ``hitmap`` columns in ``i32`` (device_row_fields(hitmap=False) then); the fallback
    container keeps the runs, a real HDF5 dataset gets the dense rows.  The per-iteration
    traces ``acceptance_rate`` / ``phids`` come from the device block's (decimated) traces when it kept them (``trace_length`` > 0;
    DeviceChains(trace_every=...)).  Not written: the wall-clock fields.  ``index`` may be in any order and hold a row once (written with one sorted fancy assignment: h5py
    wants increasing indices).  ``channel_additive`` (Tempest): the per-channel additive errors of the options file."""
    td = kind != "fdem"
    ff, fi = device_row_fields(N, K, n_depth, n_value, hitmap=hitmap and hitmap_csr is None, n_rel=n_rel, n_add=n_add, time_domain=td,
                               n_primary=n_primary, height=height, angles=angles, trace_length=trace_length)
    order = np.argsort(np.asarray(index), kind="stable")
    idx = np.asarray(index)[order]
    if idx.size > 1 and not np.all(np.diff(idx) > 0):
        raise ValueError("a sounding may be written once per call")
    f64, i32 = np.asarray(f64), np.asarray(i32)
    if not np.array_equal(order, np.arange(idx.size)):           # (rows of a line arrive in order: no copy of the hit maps then)
        f64, i32 = f64[order], i32[order]
        if hitmap_csr is not None:
            ptr0, ind0, val0 = (np.asarray(a_) for a_ in hitmap_csr)
            cnt = np.diff(ptr0)[order]
            take = np.concatenate([np.arange(ptr0[j_], ptr0[j_ + 1]) for j_ in order]) if order.size else np.zeros(0, np.int64)
            hitmap_csr = (np.r_[0, np.cumsum(cnt)], ind0[take], val0[take])
    # consecutive rows are written as a slice (a block copy; h5py likes it better, too), anything else as the sorted index vector
    sel = slice(int(idx[0]), int(idx[-1]) + 1) if idx.size and int(idx[-1]) - int(idx[0]) + 1 == idx.size else idx
    col, F, I = 0, {}, {}
    for name, w in ff:
        F[name] = f64[:, col:col + w]
        col += w
    col = 0
    for name, w in fi:
        I[name] = i32[:, col:col + w]
        col += w
    o = options
    rel_axes, add_axes = level_axes(o, kind)
    parent["iteration"][sel] = I["iterations"][:, 0]
    parent["burned_in_iteration"][sel] = np.maximum(I["burned_in_iteration"][:, 0], 0)
    parent["burned_in"][sel] = I["status"][:, 0] == 1
    parent["best_iteration"][sel] = I["best_iteration"][:, 0]
    if trace_length:
        n_tr = parent["phids/data"].shape[1]
        w = min(n_tr, trace_length)
        parent["phids/data"][sel, :w] = F["trace_misfit"][:, :w]
        parent["acceptance_rate/data"][sel, :w] = I["trace_accept"][:, :w].astype(np.uint8)
    parent["multiplier"][sel] = 1.0
    parent["halfspace/data"][sel] = np.exp(F["log_mean_prior"][:, 0])
    d = parent["data"]
    for key in ("x", "y", "z", "elevation", "line_number"):
        d[key + "/data"][sel] = F[key][:, 0]
    d["data/data"][sel, :] = F["data"]
    d["predicted_data/data"][sel, :] = F["predicted"]
    if not td:
        d["std/data"][sel, :] = np.sqrt((F["relative_error"] * F["data"]) ** 2.0 + F["additive_error"] ** 2.0)
        d["relative_error/data"][sel] = F["relative_error"][:, 0]
        d["additive_error/data"][sel] = F["additive_error"][:, 0]
        d["relative_error/posterior/values/data"][sel, :] = I["rel_hist"]
        d["additive_error/posterior/values/data"][sel, :] = I["add_hist"]
        d["relative_error/posterior/mesh/y/relative_to/data"][sel] = rel_axes[0][1]
        d["additive_error/posterior/mesh/y/relative_to/data"][sel] = add_axes[0][1]
        if height:
            d["z/data"][sel] = F["best_height"][:, 0]
            d["z/posterior/values/data"][sel, :] = I["height_hist"]
            d["z/posterior/mesh/y/relative_to/data"][sel] = F["height0"][:, 0]
    else:
        tempest = kind == "tempest"
        one = lambda a: a if a.shape[1] > 1 else a[:, 0]
        d["std/data"][sel, :] = F["std"]
        d["relative_error/data"][sel] = one(F["relative_error"])
        add_name = "additive_error_multiplier" if tempest else "additive_error"
        d[add_name + "/data"][sel] = one(F["additive_error"])
        if tempest and channel_additive is not None:
            d["additive_error/data"][sel, :] = np.broadcast_to(np.asarray(channel_additive, dtype=np.float64), (idx.size, N))
        rh, ah = I["rel_hist"].reshape(idx.size, n_rel, -1), I["add_hist"].reshape(idx.size, n_add, -1)
        for g in range(n_rel):
            d["relative_error/posterior{}/values/data".format(g)][sel, :] = rh[:, g]
            d["relative_error/posterior{}/mesh/y/relative_to/data".format(g)][sel] = rel_axes[g][1]
        for g in range(n_add):
            d[add_name + "/posterior{}/values/data".format(g)][sel, :] = ah[:, g]
            d[add_name + "/posterior{}/mesh/y/relative_to/data".format(g)][sel] = add_axes[g][1]
        # Tempest channels hold primary + secondary (Tempest_datapoint.py:106-123); the fields are stored apart as well
        nc = max(1, n_primary)
        per = N // nc
        prim = np.repeat(F["primary"], per, axis=1) if (tempest and n_primary) else 0.0
        ppri = np.repeat(F["predicted_primary"], per, axis=1) if (tempest and n_primary) else 0.0
        d["secondary_field/data"][sel, :] = F["data"] - prim
        d["predicted_secondary_field/data"][sel, :] = F["predicted"] - ppri
        if n_primary and tempest:                # (SkyTEM files carry no primary-field columns; the reference leaves those rows at their fill)
            d["primary_field/data"][sel] = one(F["primary"])
            d["predicted_primary_field/data"][sel] = one(F["predicted_primary"])
        la = np.array(F["loop_angles"], dtype=np.float64)
        off_b, z_b = np.array(F["offset"], dtype=np.float64), np.array(F["z"][:, 0], dtype=np.float64)
        cols_ = {"tx_pitch": 0, "tx_roll": 1, "tx_yaw": 2, "rx_pitch": 3, "rx_roll": 4, "rx_yaw": 5}
        for name, _ in angles:                                   # sampled scalars: the best state's instead of the file's
            if name in cols_:
                la[:, cols_[name]] = F["best_" + name][:, 0]
            elif name in ("dx", "dy", "dz"):
                off_b[:, "xyz".index(name[1])] = F["best_" + name][:, 0]
            elif name == "tx_z":
                z_b = F["best_" + name][:, 0]
        _write_loop_pair(d, sel, F["x"][:, 0], F["y"][:, 0], z_b, (off_b[:, 0], off_b[:, 1], off_b[:, 2]), la, loop_radius)
        for name, _ in angles:
            grp = _sampled_scalar_group(name)
            d[grp + "/posterior/values/data"][sel, :] = I[name + "_hist"]
            d[grp + "/posterior/mesh/y/relative_to/data"][sel] = F[name + "_centre"][:, 0]
    m = parent["model"]
    k = I["best_k"][:, 0]
    m["mesh/nCells/data"][sel] = k
    m["mesh/nCells/posterior/values/data"][sel, :] = I["k_hist"]
    # padded rows of the best model: edges 0, e_1 .. e_{k-1}, inf then NaN; values sigma_1 .. sigma_k then NaN
    jj = np.arange(K + 1)[None, :]
    kk = k.astype(np.int64)[:, None]
    rows = np.full((idx.size, K + 1), np.nan)
    rows[:, 1:K] = F["best_edges"][:, :K - 1]
    rows[:, 0] = 0.0
    rows = np.where(jj < kk, rows, np.where(jj == kk, np.inf, np.nan))
    vals = np.where(jj[:, :K] < kk, F["best_sigma"], np.nan)
    m["mesh/y/edges/data"][sel, :] = rows
    m["mesh/y/edges/posterior/values/data"][sel, :] = I["edge_hist"]
    m["values/data"][sel, :] = vals
    if hitmap and hitmap_csr is not None:
        ptr_, ind_, val_ = hitmap_csr
        target = m["values/posterior/values/data"]
        if hasattr(target, "write_run_rows"):
            target.write_run_rows(idx, ptr_, ind_, val_)
        else:                                    # a real HDF5 dataset: the dense rows
            tmp = _Dataset("tmp", (idx.size, n_value * n_depth), "i4", fillvalue=0)
            ptr0 = np.asarray(ptr_, dtype=np.int64)
            target[sel, :, :] = tmp._expand(idx.size, ptr0 - ptr0[0], np.asarray(ind_), np.asarray(val_).astype(np.int32)).reshape(idx.size, n_value, n_depth)
    elif hitmap:
        m["values/posterior/values/data"][sel, :, :] = I["hitmap"].reshape(idx.size, n_value, n_depth)
    m["values/posterior/mesh/y/relative_to/data"][sel] = F["log_mean_prior"][:, 0] / np.log(10.0)
