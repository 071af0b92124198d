"""CHECKER, run by an interpreter that has the real h5py (tests/test_hdf5_file.py finds one; /opt/conda/bin/python3.9 in this image):
    python tests/h5dump.py <file.h5> <out prefix>
opens <file.h5> with h5py and writes <out prefix>.npz (every dataset, by its path) and <out prefix>.json ({path: {"kind", "shape",
"dtype", "attrs", "attr_types", "chunks", "compression", "fillvalue"}}).  Nothing of this repository is imported."""
import json
import sys

import h5py
import numpy as np

src, out = sys.argv[1], sys.argv[2]
arrays, meta = {}, {}
with h5py.File(src, "r") as f:
    conv = lambda v: v.decode() if isinstance(v, bytes) else (v if isinstance(v, str) else np.asarray(v).tolist())

    def visit(name, obj):
        path = "/" + name
        attrs = {k: conv(v) for k, v in obj.attrs.items()}
        types = {k: type(v).__name__ for k, v in obj.attrs.items()}
        if isinstance(obj, h5py.Dataset):
            arrays[path] = obj[()]
            fv = obj.fillvalue
            meta[path] = dict(kind="dataset", shape=list(obj.shape), dtype=str(obj.dtype), attrs=attrs, attr_types=types,
                              chunks=list(obj.chunks) if obj.chunks else None, compression=obj.compression,
                              fillvalue=None if fv is None else ("nan" if (np.asarray(fv).dtype.kind == "f" and np.isnan(fv)) else np.asarray(fv).tolist()),
                              storage=int(obj.id.get_storage_size()))
        else:
            meta[path] = dict(kind="group", attrs=attrs, attr_types=types)
    meta["/"] = dict(kind="group", attrs={k: conv(v) for k, v in f.attrs.items()}, attr_types={k: type(v).__name__ for k, v in f.attrs.items()})
    f.visititems(visit)
    meta["__h5py__"] = dict(h5py=h5py.__version__, hdf5=h5py.version.hdf5_version, libver=list(f.libver))
np.savez(out + ".npz", **arrays)
json.dump(meta, open(out + ".json", "w"), sort_keys=True)
