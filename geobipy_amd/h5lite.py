"""Real HDF5 files without h5py: the few entry points of the HDF5 C library that turn a results container held in memory
(``hdf.NpzGroup``: groups, datasets, attributes) into the file h5py would have written for the same calls.

Why: the reference's post-processing opens a flight line's results with ``h5py.File`` (inversion/Inference2D.py:328-331).  h5py is a
CPython extension and cannot be installed into this interpreter, but the HDF5 library itself is a plain C library, and many images
ship one (here: ``/opt/conda/lib/libhdf5.so``, 1.10.6, next to a Python 3.9 that has h5py -- which tests/test_hdf5_file.py uses as
the CHECKER: it opens what this module wrote with the real h5py).  ``ctypes`` is all the binding there is.

What is written, and how h5py would have done it (so that a reader cannot tell):
  * groups by ``H5Gcreate2``; scalar datasets on a scalar dataspace, arrays on a simple one; native little-endian types;
  * ``numpy.bool_`` as h5py's enum {FALSE = 0, TRUE = 1} over int8;
  * ``str`` attributes as variable-length UTF-8 strings on a scalar dataspace (``obj.attrs[k] = "text"`` in h5py);
  * a dataset nothing was written to: created with its fill value, no storage allocated -- reads return the fill value;
  * dense datasets contiguous and uncompressed (h5py's default; the reference's own files are exactly that); the large ones (misfit
    traces) are allocated early and their bytes written at the address ``H5Dget_offset`` reports once the file is closed, outside the lock;
  * a dataset held as runs (the conductivity-depth hit maps, hdf._Dataset.write_run_rows): chunked one sounding per chunk with the
    deflate filter; each chunk's zlib stream is made by the calling thread -- written straight from the runs by the package's native
    encoder (gbp_runs_to_zlib: O(runs), ~50 us per sounding against ~1 ms for zlib over the 440 KB dense row) -- and handed over with
    ``H5Dwrite_chunk`` (the library only files the bytes); rows nothing was written to have no chunk.  The reference stores these maps
    dense and uncompressed -- 440 KB per sounding; h5py reads either transparently.
The library is not assumed thread-safe: every call into it is made under one lock; compression happens outside it.
"""
import ctypes
import ctypes.util
import glob
import os
import threading
import zlib

import numpy as np

_LOCK = threading.RLock()
_lib = None
_err = None

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int
H5F_ACC_TRUNC = 0x0002
H5S_SCALAR = 0
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5Z_FILTER_DEFLATE = 1
H5D_ALLOC_TIME_EARLY = 1
H5D_FILL_TIME_NEVER = 1
HADDR_UNDEF = ctypes.c_uint64(-1).value
# Dense datasets of at least this many bytes are not copied by H5Dwrite under the library's lock: their contiguous storage is allocated
# when the dataset is created, its address taken (H5Dget_offset), and the bytes are written into the closed file at that address by the
# calling thread (os.pwrite) -- sixteen flight lines' 16 MB misfit traces then land side by side instead of one after the other.
DIRECT_WRITE_BYTES = 1 << 20


def _candidates():
    if os.environ.get("GBP_LIBHDF5"):
        yield os.environ["GBP_LIBHDF5"]
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _sig(lib, name, res, *args):
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, list(args)
    return fn


def load():
    """The HDF5 library (>= 1.10.3: 64-bit identifiers, ``H5Dwrite_chunk``, the deflate filter) or None; the reason is kept in ``why_not()``."""
    global _lib, _err
    if _lib is not None or _err is not None:
        return _lib
    with _LOCK:
        if _lib is not None or _err is not None:
            return _lib
        if os.environ.get("GBP_LIBHDF5") == "none":
            _err = "disabled by GBP_LIBHDF5=none"
            return None
        tried = []
        for path in _candidates():
            try:
                lib = ctypes.CDLL(path)
                maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
                if lib.H5open() < 0 or lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel)) < 0:
                    raise OSError("H5open failed")
                ver = (maj.value, mnr.value, rel.value)
                if ver < (1, 10, 3):
                    raise OSError("version %d.%d.%d < 1.10.3" % ver)
                _sig(lib, "H5Zfilter_avail", ctypes.c_int, ctypes.c_int)
                if lib.H5Zfilter_avail(H5Z_FILTER_DEFLATE) <= 0:
                    raise OSError("built without the deflate filter")
                _bind(lib)
                lib.path, lib.version = path, ver
                _lib = lib
                return lib
            except (OSError, AttributeError) as e:
                tried.append("{}: {}".format(path, e))
        _err = "; ".join(tried) if tried else "no libhdf5 found (GBP_LIBHDF5=<path> names one)"
        return None


def why_not():
    load()
    return _err


def available():
    return load() is not None


def _bind(lib):
    P, c_int, c_uint, c_size_t, c_char_p, c_void_p = ctypes.POINTER, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p
    _sig(lib, "H5Eset_auto2", herr_t, hid_t, c_void_p, c_void_p)
    _sig(lib, "H5Fcreate", hid_t, c_char_p, c_uint, hid_t, hid_t)
    _sig(lib, "H5Fclose", herr_t, hid_t)
    _sig(lib, "H5Gcreate2", hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t)
    _sig(lib, "H5Gclose", herr_t, hid_t)
    _sig(lib, "H5Screate", hid_t, c_int)
    _sig(lib, "H5Screate_simple", hid_t, c_int, P(hsize_t), P(hsize_t))
    _sig(lib, "H5Sclose", herr_t, hid_t)
    _sig(lib, "H5Dcreate2", hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    _sig(lib, "H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    _sig(lib, "H5Dwrite_chunk", herr_t, hid_t, hid_t, ctypes.c_uint32, P(hsize_t), c_size_t, c_void_p)
    _sig(lib, "H5Dclose", herr_t, hid_t)
    _sig(lib, "H5Dget_offset", ctypes.c_uint64, hid_t)
    _sig(lib, "H5Pset_alloc_time", herr_t, hid_t, c_int)
    _sig(lib, "H5Pset_fill_time", herr_t, hid_t, c_int)
    _sig(lib, "H5Pcreate", hid_t, hid_t)
    _sig(lib, "H5Pset_chunk", herr_t, hid_t, c_int, P(hsize_t))
    _sig(lib, "H5Pset_deflate", herr_t, hid_t, c_uint)
    _sig(lib, "H5Pset_fill_value", herr_t, hid_t, hid_t, c_void_p)
    _sig(lib, "H5Pclose", herr_t, hid_t)
    _sig(lib, "H5Acreate2", hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t)
    _sig(lib, "H5Awrite", herr_t, hid_t, hid_t, c_void_p)
    _sig(lib, "H5Aclose", herr_t, hid_t)
    _sig(lib, "H5Tcopy", hid_t, hid_t)
    _sig(lib, "H5Tset_size", herr_t, hid_t, c_size_t)
    _sig(lib, "H5Tset_cset", herr_t, hid_t, c_int)
    _sig(lib, "H5Tenum_create", hid_t, hid_t)
    _sig(lib, "H5Tenum_insert", herr_t, hid_t, c_char_p, c_void_p)
    _sig(lib, "H5Tclose", herr_t, hid_t)
    # the reader's entry points
    _sig(lib, "H5Fopen", hid_t, c_char_p, c_uint, hid_t)
    _sig(lib, "H5Oopen", hid_t, hid_t, c_char_p, hid_t)
    _sig(lib, "H5Oclose", herr_t, hid_t)
    _sig(lib, "H5Iget_type", c_int, hid_t)
    _sig(lib, "H5Dget_space", hid_t, hid_t)
    _sig(lib, "H5Dget_type", hid_t, hid_t)
    _sig(lib, "H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    _sig(lib, "H5Sget_simple_extent_ndims", c_int, hid_t)
    _sig(lib, "H5Sget_simple_extent_dims", c_int, hid_t, P(hsize_t), P(hsize_t))
    _sig(lib, "H5Sget_simple_extent_npoints", ctypes.c_int64, hid_t)
    _sig(lib, "H5Tget_class", c_int, hid_t)
    _sig(lib, "H5Tget_size", c_size_t, hid_t)
    _sig(lib, "H5Tget_sign", c_int, hid_t)
    _sig(lib, "H5Tis_variable_str", c_int, hid_t)
    _sig(lib, "H5Tget_nmembers", c_int, hid_t)
    _sig(lib, "H5Aget_num_attrs", c_int, hid_t)
    _sig(lib, "H5Aopen_by_idx", hid_t, hid_t, c_char_p, c_int, c_int, hsize_t, hid_t, hid_t)
    _sig(lib, "H5Aget_name", ctypes.c_ssize_t, hid_t, c_size_t, c_char_p)
    _sig(lib, "H5Aget_type", hid_t, hid_t)
    _sig(lib, "H5Aget_space", hid_t, hid_t)
    _sig(lib, "H5Aread", herr_t, hid_t, hid_t, c_void_p)
    _sig(lib, "H5Dvlen_reclaim", herr_t, hid_t, hid_t, hid_t, c_void_p)
    lib.iter_cb = ctypes.CFUNCTYPE(herr_t, hid_t, c_char_p, c_void_p, c_void_p)
    _sig(lib, "H5Literate", herr_t, hid_t, c_int, c_int, P(hsize_t), lib.iter_cb, c_void_p)
    g = lambda n: hid_t.in_dll(lib, n).value
    lib.t = {np.dtype(k): g(v) for k, v in {
        "f8": "H5T_NATIVE_DOUBLE_g", "f4": "H5T_NATIVE_FLOAT_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
        "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g",
        "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g"}.items()}
    lib.dcpl_class = g("H5P_CLS_DATASET_CREATE_ID_g")
    lib.c_s1 = g("H5T_C_S1_g")
    # h5py's boolean: an enum over int8 (h5py/h5t.pyx: _b8 / "FALSE", "TRUE")
    b = lib.H5Tenum_create(lib.t[np.dtype("i1")])
    for name, val in ((b"FALSE", 0), (b"TRUE", 1)):
        v = ctypes.c_int8(val)
        if lib.H5Tenum_insert(b, name, ctypes.byref(v)) < 0:
            raise OSError("H5Tenum_insert failed")
    lib.t[np.dtype("bool")] = b
    s = lib.H5Tcopy(lib.c_s1)
    if s < 0 or lib.H5Tset_size(s, H5T_VARIABLE) < 0 or lib.H5Tset_cset(s, H5T_CSET_UTF8) < 0:
        raise OSError("variable-length string type failed")
    lib.vlen_str = s
    lib.H5Eset_auto2(0, None, None)               # errors are raised here, not printed by the library


class HDF5Error(OSError):
    pass


def _ok(v, what):
    if v < 0:
        raise HDF5Error("HDF5: {} failed".format(what))
    return v


def _dims(shape):
    return (hsize_t * len(shape))(*[int(s) for s in shape])


def _space(lib, shape):
    if len(shape) == 0:
        return _ok(lib.H5Screate(H5S_SCALAR), "H5Screate")
    return _ok(lib.H5Screate_simple(len(shape), _dims(shape), None), "H5Screate_simple")


def _write_attrs(lib, obj, attrs):
    """h5py's ``obj.attrs[k] = v`` for the values the containers carry: a str (scalar variable-length UTF-8 string), a list of str (1-D
    array of such strings: a time-domain system's .stm lines, TdemSystem_GAAEM.py:118), numeric scalars and arrays."""
    enc = lambda x: x.encode("utf-8") if isinstance(x, str) else bytes(x)
    for k in sorted(attrs):
        v = attrs[k]
        strings = None
        if isinstance(v, (str, bytes)):
            strings, shape = [enc(v)], ()
        elif isinstance(v, (list, tuple)) and len(v) > 0 and all(isinstance(x, (str, bytes)) for x in v):
            strings, shape = [enc(x) for x in v], (len(v),)
        if strings is not None:
            space = _space(lib, shape)
            try:
                a = _ok(lib.H5Acreate2(obj, k.encode(), lib.vlen_str, space, 0, 0), "H5Acreate2 " + k)
                buf = (ctypes.c_char_p * len(strings))(*strings)
                _ok(lib.H5Awrite(a, lib.vlen_str, buf), "H5Awrite " + k)
                lib.H5Aclose(a)
            finally:
                lib.H5Sclose(space)
            continue
        arr = np.asarray(v)
        if arr.dtype not in lib.t:
            raise HDF5Error("attribute {!r}: only strings, lists of strings and numeric values are written ({!r})".format(k, v))
        space = _space(lib, arr.shape)
        try:
            a = _ok(lib.H5Acreate2(obj, k.encode(), lib.t[arr.dtype], space, 0, 0), "H5Acreate2 " + k)
            tmp = np.ascontiguousarray(arr)
            if tmp.dtype == np.bool_:
                tmp = tmp.view(np.int8)
            _ok(lib.H5Awrite(a, lib.t[arr.dtype], tmp.ctypes.data_as(ctypes.c_void_p)), "H5Awrite " + k)
            lib.H5Aclose(a)
        finally:
            lib.H5Sclose(space)


def _fill_plist(lib, dtype, fill):
    """Dataset-creation property list with the fill value (None: the library's default, zero)."""
    dcpl = _ok(lib.H5Pcreate(lib.dcpl_class), "H5Pcreate")
    if fill is not None:
        f = np.array(fill, dtype=dtype)
        _ok(lib.H5Pset_fill_value(dcpl, lib.t[np.dtype(dtype)], f.ctypes.data_as(ctypes.c_void_p)), "H5Pset_fill_value")
    return dcpl


def _deflated_rows(ds, level):
    """[(row, zlib stream of the row's dense chunk)] of a dataset held as runs -- the expensive part, outside the library's lock.
    int32 maps go through the package's native encoder, which writes the stream FROM the runs (gbp_runs_to_zlib, csrc/gbp_hostpack.h:
    ~50 us per sounding); anything else is expanded and handed to zlib (~1 ms per 440 KB row)."""
    rows, ptr, start, val = ds.runs()
    M = int(np.prod(ds.shape[1:]))
    if np.dtype(ds.dtype) == np.int32 and rows.size:
        from . import _lib
        try:
            lib = _lib.load()
        except _lib.NativeLibraryError:
            lib = None
        if lib is not None:
            ptr64, st, vl = np.ascontiguousarray(ptr, dtype=np.int64), np.ascontiguousarray(start, dtype=np.int32), np.ascontiguousarray(val, dtype=np.int32)
            cap = 16 * st.size + rows.size * (M // 32 + 64)
            buf, optr = np.empty(cap, dtype=np.uint8), np.empty(rows.size + 1, dtype=np.int64)
            _lib.check(lib.gbp_runs_to_zlib(int(rows.size), M, ptr64.ctypes.data, st.ctypes.data, vl.ctypes.data, buf.ctypes.data, cap, optr.ctypes.data))
            whole = buf[:optr[-1]].tobytes()
            return [(int(rows[j]), whole[optr[j]:optr[j + 1]]) for j in range(rows.size)]
    out = []
    for j in range(rows.size):
        a, b = int(ptr[j]), int(ptr[j + 1])
        dense = np.repeat(val[a:b], np.diff(np.r_[start[a:b].astype(np.int64), M]))
        assert dense.size == M
        out.append((int(rows[j]), zlib.compress(dense.astype(ds.dtype, copy=False).tobytes(), level)))
    return out


def write_tree(path, root, compresslevel=1):
    """``root`` (an ``hdf.NpzGroup``) -> the HDF5 file ``path``.  Returns the path."""
    lib = load()
    if lib is None:
        raise HDF5Error("no usable HDF5 library: " + str(_err))
    from .hdf import NpzGroup
    # 1. everything that takes time and needs no library call: the deflated chunks of the datasets held as runs
    chunks = {}

    def prepare(g):
        for v in g._items.values():
            if isinstance(v, NpzGroup):
                prepare(v)
            elif v.sparse:
                chunks[v.name] = _deflated_rows(v, compresslevel)
            elif v.dtype not in lib.t:
                raise HDF5Error("dataset {}: dtype {} is not written".format(v.name, v.dtype))
    prepare(root)

    direct = []                                   # (file address, array) of the large dense datasets, written after the file is closed

    # 2. the file, under the lock
    def emit(loc, g):
        _write_attrs(lib, loc, g.attrs)
        for name in sorted(g._items):
            v = g._items[name]
            if isinstance(v, NpzGroup):
                sub = _ok(lib.H5Gcreate2(loc, name.encode(), 0, 0, 0), "H5Gcreate2 " + v.name)
                try:
                    emit(sub, v)
                finally:
                    lib.H5Gclose(sub)
                continue
            ftype = lib.t[np.dtype(v.dtype)]
            space = _space(lib, v.shape)
            dcpl = _fill_plist(lib, v.dtype, v._fill)
            try:
                a = None
                if v.name in chunks:
                    cdims = (1,) + tuple(v.shape[1:])
                    _ok(lib.H5Pset_chunk(dcpl, len(cdims), _dims(cdims)), "H5Pset_chunk")
                    _ok(lib.H5Pset_deflate(dcpl, compresslevel), "H5Pset_deflate")
                elif v.materialised:
                    a = np.ascontiguousarray(v.arr)
                    if a.dtype == np.bool_:
                        a = a.view(np.int8)
                    if a.nbytes >= DIRECT_WRITE_BYTES:
                        _ok(lib.H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY), "H5Pset_alloc_time")
                        _ok(lib.H5Pset_fill_time(dcpl, H5D_FILL_TIME_NEVER), "H5Pset_fill_time")
                d = _ok(lib.H5Dcreate2(loc, name.encode(), ftype, space, 0, dcpl, 0), "H5Dcreate2 " + v.name)
                try:
                    if v.name in chunks:
                        off = (hsize_t * len(v.shape))()
                        for row, blob in chunks[v.name]:
                            off[0] = row
                            _ok(lib.H5Dwrite_chunk(d, 0, 0, off, len(blob), blob), "H5Dwrite_chunk " + v.name)
                    elif a is not None and a.size:
                        addr = lib.H5Dget_offset(d) if a.nbytes >= DIRECT_WRITE_BYTES else HADDR_UNDEF
                        if addr != HADDR_UNDEF:
                            direct.append((int(addr), a))
                        else:
                            _ok(lib.H5Dwrite(d, ftype, 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)), "H5Dwrite " + v.name)
                    _write_attrs(lib, d, v.attrs)
                finally:
                    lib.H5Dclose(d)
            finally:
                lib.H5Pclose(dcpl)
                lib.H5Sclose(space)

    # The file is built under a temporary name and renamed when it is complete: the large datasets are created with fill time NEVER and
    # receive their bytes AFTER the library has closed the file, so an exception in between must not leave a structurally valid
    # container whose large datasets hold whatever the blocks held.  (The file is created with the default creation properties -- no
    # user block --, so H5Dget_offset is the absolute file address; an undefined address falls back to H5Dwrite above.)
    tmp = "{}.tmp{}".format(path, os.getpid())
    try:
        with _LOCK:
            f = _ok(lib.H5Fcreate(os.fsencode(tmp), H5F_ACC_TRUNC, 0, 0), "H5Fcreate " + tmp)
            try:
                emit(f, root)
            finally:
                _ok(lib.H5Fclose(f), "H5Fclose")
        if direct:                                # outside the lock: the raw bytes of the large datasets, at the addresses the library gave
            fd = os.open(tmp, os.O_WRONLY)
            try:
                size = os.fstat(fd).st_size
                for addr, a in direct:
                    if addr + a.nbytes > size:    # (early allocation reserves the block inside the file the library closed)
                        raise HDF5Error("dataset block at {} + {} bytes lies outside the {}-byte file".format(addr, a.nbytes, size))
                    view, done = memoryview(a.reshape(-1).view(np.uint8)), 0
                    while done < len(view):
                        done += os.pwrite(fd, view[done:], addr + done)
            finally:
                os.close(fd)
        os.replace(tmp, str(path))
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise
    return str(path)


# ---------------------------------------------------------------------------------------------------------------------------------
# Reading: a results container -- this package's or the reference's own -- without h5py
# ---------------------------------------------------------------------------------------------------------------------------------
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8


def _numpy_type(lib, tid):
    """(numpy dtype, memory type id to read with, is variable-length string) of an HDF5 datatype; None for what is not handled."""
    cls, size = lib.H5Tget_class(tid), int(lib.H5Tget_size(tid))
    if cls == H5T_FLOAT and size in (4, 8):
        dt = np.dtype("f%d" % size)
        return dt, lib.t[dt], False
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        dt = np.dtype(("i" if lib.H5Tget_sign(tid) == 1 else "u") + str(size))
        return dt, lib.t[dt], False
    if cls == H5T_ENUM and size == 1 and lib.H5Tget_nmembers(tid) == 2:       # h5py's boolean
        return np.dtype("bool"), lib.t[np.dtype("bool")], False
    if cls == H5T_STRING and lib.H5Tis_variable_str(tid) > 0:
        return np.dtype("O"), lib.vlen_str, True
    return None, None, False


def _shape_of(lib, space):
    nd = lib.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (hsize_t * nd)()
    lib.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _read_attrs(lib, obj):
    out = {}
    for n in range(max(0, lib.H5Aget_num_attrs(obj))):
        a = _ok(lib.H5Aopen_by_idx(obj, b".", 0, 0, n, 0, 0), "H5Aopen_by_idx")
        try:
            ln = lib.H5Aget_name(a, 0, None)
            buf = ctypes.create_string_buffer(ln + 1)
            lib.H5Aget_name(a, ln + 1, buf)
            tid, space = lib.H5Aget_type(a), lib.H5Aget_space(a)
            try:
                dt, mem, vlen = _numpy_type(lib, tid)
                shape = _shape_of(lib, space)
                n_el = int(np.prod(shape)) if shape else 1
                if dt is None:
                    continue
                if vlen:
                    ptrs = (ctypes.c_char_p * n_el)()
                    _ok(lib.H5Aread(a, mem, ptrs), "H5Aread")
                    vals = [(p_ or b"").decode("utf-8") for p_ in ptrs]
                    lib.H5Dvlen_reclaim(mem, space, 0, ptrs)
                    out[buf.value.decode()] = vals[0] if shape == () else vals
                else:
                    arr = np.empty(shape, dtype=np.int8 if dt == np.bool_ else dt)
                    _ok(lib.H5Aread(a, mem, arr.ctypes.data_as(ctypes.c_void_p)), "H5Aread")
                    arr = arr.astype(bool) if dt == np.bool_ else arr
                    out[buf.value.decode()] = arr.item() if shape == () else arr
            finally:
                lib.H5Tclose(tid)
                lib.H5Sclose(space)
        finally:
            lib.H5Aclose(a)
    return out


def read_tree(path):
    """({hdf path: array} of every dataset, {hdf path: attributes} of every group and dataset that has any) of an HDF5 file -- a results
    container written by this package or by the reference (numeric, boolean and variable-length string data: what those files hold)."""
    lib = load()
    if lib is None:
        raise HDF5Error("no usable HDF5 library: " + str(_err))
    arrays, attrs = {}, {}

    def children(gid):
        names = []
        cb = lib.iter_cb(lambda g_, name, info, data: names.append(name) or 0)
        idx = hsize_t(0)
        _ok(lib.H5Literate(gid, 0, 0, ctypes.byref(idx), cb, None), "H5Literate")
        return names

    def walk(gid, prefix):
        a = _read_attrs(lib, gid)
        if a:
            attrs[prefix or "/"] = a
        for name in children(gid):
            obj = _ok(lib.H5Oopen(gid, name, 0), "H5Oopen")
            try:
                path_ = prefix + "/" + name.decode()
                kind = lib.H5Iget_type(obj)
                if kind == H5I_GROUP:
                    walk(obj, path_)
                elif kind == H5I_DATASET:
                    tid, space = lib.H5Dget_type(obj), lib.H5Dget_space(obj)
                    try:
                        dt, mem, vlen = _numpy_type(lib, tid)
                        shape = _shape_of(lib, space)
                        if dt is not None and not vlen:
                            arr = np.empty(shape, dtype=np.int8 if dt == np.bool_ else dt)
                            if arr.size:
                                _ok(lib.H5Dread(obj, mem, 0, 0, 0, arr.ctypes.data_as(ctypes.c_void_p)), "H5Dread " + path_)
                            arrays[path_] = arr.astype(bool) if dt == np.bool_ else arr
                    finally:
                        lib.H5Tclose(tid)
                        lib.H5Sclose(space)
                    a2 = _read_attrs(lib, obj)
                    if a2:
                        attrs[path_] = a2
            finally:
                lib.H5Oclose(obj)

    with _LOCK:
        f = _ok(lib.H5Fopen(os.fsencode(str(path)), 0, 0), "H5Fopen " + str(path))
        try:
            root = _ok(lib.H5Oopen(f, b"/", 0), "H5Oopen /")
            try:
                walk(root, "")
            finally:
                lib.H5Oclose(root)
        finally:
            lib.H5Fclose(f)
    return arrays, attrs
