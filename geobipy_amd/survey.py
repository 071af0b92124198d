"""Survey-level entry (reduced to what feeds the path): the reference's user inputs -- an options file
and an FDEM or TDEM data CSV -- to the posteriors of every sounding, on the GPUs of one node.

Mirrors what the reference's harness does around ``Inference1D`` without its control plane:
  ``read_options``      inversion/user_parameters.py:31-99 (same keys, same defaults, same required keys)
  ``FdemData.read_csv`` classes/data/dataset/FdemData.py:620-682 + Data.py:505-528 + pointcloud/Point.py:355-382
                        (column recognition by header name), one row per sounding
  ``TdemData.read_csv`` classes/data/dataset/TdemData.py (columns S<system><component>_time_<t>, txrx_d{x,y,z})
  ``infer``             inversion/Inference3D.py:518-635: soundings are independent; a rank owns one GPU and a
                        contiguous block of soundings (base/MPI.py:172-201) and runs all of its chains in lockstep
                        (``DeviceChains`` under the reference's burn-in / stop schedule); results are gathered on rank 0.
The reference's HDF5 output (row f-4) is not reproduced: results are numpy arrays (``SurveyResult.save`` -> .npz).
"""
import ast
import os

import numpy as np

from .datapoint import FdemDataPoint
from .system import FdemSystem

REQUIRED_KEYS = ("data_type", "data_filename", "system_filename", "n_markov_chains", "interactive_plot", "update_plot_every",
                 "save_png", "save_hdf5", "solve_parameter", "solve_gradient", "maximum_number_of_layers", "minimum_depth",
                 "maximum_depth", "probability_of_birth", "probability_of_death", "probability_of_perturb",
                 "probability_of_no_change")
_DATA_TYPES = ("FdemData", "TdemData", "TempestData", "FdemDataPoint", "TdemDataPoint", "TempestDataPoint")


def _value(node):
    """Literals, arithmetic on literals, lists / tuples, and the data-type names (kept as strings).  The reference
    exec()s the file (user_parameters.py:83-88); the files it ships need no more than this."""
    if isinstance(node, ast.Name):
        if node.id in _DATA_TYPES:
            return node.id
        if node.id in ("inf", "nan"):
            return float(node.id)
        raise ValueError("unknown name {!r} in options file".format(node.id))
    if isinstance(node, (ast.List, ast.Tuple)):
        return [_value(e) for e in node.elts]
    if (isinstance(node, ast.Subscript) and isinstance(node.value, ast.Attribute) and node.value.attr == "r_"
            and isinstance(node.value.value, ast.Name) and node.value.value.id in ("np", "numpy")):
        # np.r_[a, b, ...] of the reference's tempest_options (per-channel additive errors): a flat list of numbers
        sl = node.slice
        parts = [_value(e) for e in sl.elts] if isinstance(sl, ast.Tuple) else [_value(sl)]
        return [float(x) for q in parts for x in (q if isinstance(q, list) else [q])]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _value(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp):
        a, b = _value(node.left), _value(node.right)
        ops = {ast.Add: lambda: a + b, ast.Sub: lambda: a - b, ast.Mult: lambda: a * b, ast.Div: lambda: a / b,
               ast.Pow: lambda: a ** b}
        if type(node.op) not in ops:
            raise ValueError("unsupported operator in options file")
        return ops[type(node.op)]()
    return ast.literal_eval(node)


def read_options(filename, **overrides):
    """dict of the options file's assignments with the reference's defaults filled in and the file names joined to
    ``data_directory`` (relative directories are taken relative to the options file)."""
    with open(filename) as f:
        tree = ast.parse(f.read(), filename=filename)
    o = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            o[node.targets[0].id] = _value(node.value)
    o.update({k: v for k, v in overrides.items() if v is not None})
    missing = [k for k in REQUIRED_KEYS if k not in o]
    if missing:
        raise ValueError("Missing {} from the user parameter file".format(missing))
    for key, default in (("gradient_standard_deviation", 1.5), ("multiplier", 1.0), ("factor", 10.0), ("covariance_scaling", 1.0)):
        if o.get(key) is None:
            o[key] = default
    o["stochastic_newton"] = not o.get("ignore_likelihood", False)
    base = o.get("data_directory", "")
    if not os.path.isabs(base):
        base = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(filename)), base))
    join = lambda v: [os.path.join(base, x) for x in v] if isinstance(v, list) else os.path.join(base, v)
    o["data_filename"], o["system_filename"] = join(o["data_filename"]), join(o["system_filename"])
    return o


class FdemData:
    """A set of FDEM soundings (classes/data/dataset/FdemData.py): per-sounding location, altitude and the 2 F data
    channels [in-phase F | quadrature F] in ppm, optional standard deviations."""

    def __init__(self, system, lineNumber, fiducial, x, y, z, elevation, data, std=None):
        self.system = system if isinstance(system, FdemSystem) else FdemSystem.read(system)
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.lineNumber, self.fiducial = f64(lineNumber), f64(fiducial)
        self.x, self.y, self.z, self.elevation = f64(x), f64(y), f64(z), f64(elevation)
        self.data = f64(data)
        self.std = None if std is None else f64(std)
        assert self.data.shape == (self.nPoints, 2 * self.system.nFrequencies), ValueError(
            "data must have shape (nPoints, 2 * nFrequencies)")

    @property
    def nPoints(self):
        return self.x.size

    @property
    def nChannels(self):
        return self.data.shape[1]

    @staticmethod
    def _csv_channels(header):
        """Column roles by header name, the reference's rules (case-insensitive)."""
        roles = dict(line=("line", "linenumber", "line_number"), fid=("fid", "fiducial", "id"), x=("e", "x", "easting"),
                     y=("n", "y", "northing"), z=("alt", "altitude", "laser", "bheight", "height"),
                     elev=("dtm", "dem_elev", "dem_np", "topo", "elev", "elevation"))
        loc, inphase, quad, in_err, quad_err = {}, [], [], [], []
        for j, name in enumerate(header):
            c = name.strip().lower()
            role = next((r for r, names in roles.items() if c in names), None)
            if role is not None:
                loc[role] = j
            elif any(label in c for label in ("cpi", "i_", "in_phase")):
                (in_err if "err" in c else inphase).append(j)
            elif any(label in c for label in ("cpq", "q_", "quad")):
                (quad_err if "err" in c else quad).append(j)
        if "line" not in loc or "fid" not in loc:
            raise ValueError("File must contain columns for line and fiducial")
        if not all(r in loc for r in ("x", "y", "z")):
            raise ValueError("File must contain columns for easting, northing, height. May also have an elevation column")
        return loc, inphase + quad, in_err + quad_err

    @classmethod
    def read_csv(cls, data_filename, system_filename):
        system = system_filename if isinstance(system_filename, FdemSystem) else FdemSystem.read(system_filename)
        with open(data_filename) as f:
            first = f.readline()
        sep = "," if "," in first else None
        header = [h for h in (first.strip().split(sep) if sep else first.split())]
        loc, dcols, ecols = cls._csv_channels(header)
        if len(dcols) != 2 * system.nFrequencies:
            raise ValueError("Number of data columns {} in {} does not match 2 * nFrequencies {}".format(
                len(dcols), data_filename, 2 * system.nFrequencies))
        table = np.atleast_2d(np.loadtxt(data_filename, delimiter=sep, skiprows=1))
        col = lambda r: table[:, loc[r]]
        elev = col("elev") if "elev" in loc else np.zeros(table.shape[0])
        std = table[:, ecols] if len(ecols) == len(dcols) else None
        return cls(system, col("line"), col("fid"), col("x"), col("y"), col("z"), elev, table[:, dcols], std)

    def subset(self, rows):
        return FdemData(self.system, self.lineNumber[rows], self.fiducial[rows], self.x[rows], self.y[rows], self.z[rows],
                        self.elevation[rows], self.data[rows], None if self.std is None else self.std[rows])

    def datapoint(self, i):
        """Sounding i as the per-sounding object (FdemData.datapoint, FdemData.py:447-487)."""
        return FdemDataPoint(x=self.x[i], y=self.y[i], z=self.z[i], elevation=self.elevation[i], data=self.data[i],
                             std=None if self.std is None else self.std[i], system=self.system,
                             lineNumber=self.lineNumber[i], fiducial=self.fiducial[i])


class TdemData:
    """A set of time-domain soundings (classes/data/dataset/TdemData.py): location, altitude, transmitter-receiver offset and
    the window data of every system, columns ``S<system><component>_time_<t>`` of the reference's CSV files in file order
    (system 0 component X then Z windows, system 1 ...), which is TdemBatch's channel layout.  ``offset``: one (dx, dy, dz) for
    the set or one per sounding [nPoints, 3] (columns txrx_dx / dy / dz); the Hankel tables depend on it, so ``infer`` builds one
    table set per distinct (horizontal distance, dz) and every chain runs with its own.  ``primary_field`` [nPoints, components]:
    the PX / PY / PZ columns of Tempest files (TempestData.py), kept for the caller, NaN when absent.  ``loop_angles``
    [nPoints, 6]: the file's tx_pitch, tx_roll, tx_yaw, rx_pitch, rx_roll, rx_yaw columns (degrees, the reference's own
    convention; TdemData.py:588-589); ``attitude`` is what Loop_pair.Geometry hands GA-AEM from them (roll, -pitch, -yaw per
    loop, Loop_pair.py:70-77) and what the kernels' geometry mixing takes."""

    MAX_OFFSET_SETS = 4096

    def __init__(self, system, lineNumber, fiducial, x, y, z, elevation, data, offset, primary_field=None, loop_angles=None):
        from .tdem import TdemSystem
        systems = [system] if isinstance(system, (str, TdemSystem)) else list(system)
        self.system = [s if isinstance(s, TdemSystem) else TdemSystem(s) for s in systems]
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.lineNumber, self.fiducial = f64(lineNumber), f64(fiducial)
        self.x, self.y, self.z, self.elevation = f64(x), f64(y), f64(z), f64(elevation)
        self.data = f64(data)
        off = np.asarray(offset, dtype=np.float64)
        self.offsets = np.array(np.broadcast_to(off, (self.x.size, 3)), dtype=np.float64)      # per sounding (own, writable copy)
        self.primary_field = None if primary_field is None else f64(primary_field)
        self.loop_angles = np.zeros((self.x.size, 6)) if loop_angles is None else np.array(np.broadcast_to(f64(loop_angles), (self.x.size, 6)))
        n = sum(s.n_components * s.nwindows for s in self.system)
        assert self.data.shape == (self.nPoints, n), ValueError("data must have shape (nPoints, {})".format(n))

    @property
    def nPoints(self):
        return self.x.size

    @property
    def nChannels(self):
        return self.data.shape[1]

    @property
    def attitude(self):
        """[nPoints, 6] (tx roll, pitch, yaw, rx roll, pitch, yaw) in GA-AEM's convention (Loop_pair.py:70-77)."""
        a = self.loop_angles
        return np.stack([a[:, 1], -a[:, 0], -a[:, 2], a[:, 4], -a[:, 3], -a[:, 5]], axis=1)

    @property
    def offset(self):
        """The one transmitter-receiver offset of the set (raises when the soundings do not share one)."""
        assert np.all(self.offsets == self.offsets[0]), ValueError("the soundings have different transmitter-receiver offsets; use .offsets")
        return tuple(float(v) for v in self.offsets[0])

    def offset_groups(self, rows=None):
        """[(offset triple, row indices)] of the distinct offsets among ``rows`` (default: all), in order of first appearance."""
        rows = np.arange(self.nPoints) if rows is None else np.asarray(rows)
        uniq, first, inverse = np.unique(self.offsets[rows], axis=0, return_index=True, return_inverse=True)
        order = np.argsort(first)
        return [(tuple(float(v) for v in uniq[g]), rows[np.nonzero(inverse.ravel() == g)[0]]) for g in order]

    @classmethod
    def read_csv(cls, data_filename, system_filename):
        with open(data_filename) as f:
            header = [h.strip() for h in f.readline().strip().split(",")]
        low = [h.lower() for h in header]
        find = lambda names: next(j for j, h in enumerate(low) if h in names)
        roles = dict(line=("line", "linenumber", "line_number"), fid=("fid", "fiducial", "id"), x=("e", "x", "easting"),
                     y=("n", "y", "northing"), z=("alt", "altitude", "laser", "bheight", "height"))
        idx = {r: find(names) for r, names in roles.items()}
        elev = next((j for j, h in enumerate(low) if h in ("dtm", "dem_elev", "dem_np", "topo", "elev", "elevation")), None)
        # data columns: the reference's rule (TdemData.py:622-631): a header containing off_time / x_time / y_time / z_time is a
        # window of the secondary field, unless it also contains 'err' (an error column, not read here)
        is_win = lambda h: any(t in h for t in ("off_time", "x_time", "y_time", "z_time"))
        dcols = [j for j, h in enumerate(low) if is_win(h) and "err" not in h]
        if not dcols:
            raise ValueError("{}: no data columns (headers containing off_time, x_time, y_time or z_time); found {}".format(
                data_filename, header))
        table = np.atleast_2d(np.loadtxt(data_filename, delimiter=",", skiprows=1))
        off = np.stack([table[:, low.index(k)] for k in ("txrx_dx", "txrx_dy", "txrx_dz")], axis=1)
        pcols = sorted(j for j, h in enumerate(low) if h in ("px", "py", "pz"))            # TdemData.py:632, primary_channels.sort()
        angles = np.stack([table[:, low.index(k)] if k in low else np.zeros(table.shape[0])
                           for k in ("tx_pitch", "tx_roll", "tx_yaw", "rx_pitch", "rx_roll", "rx_yaw")], axis=1)
        c = lambda r: table[:, idx[r]]
        return cls(system_filename, c("line"), c("fid"), c("x"), c("y"), c("z"),
                   table[:, elev] if elev is not None else np.zeros(table.shape[0]), table[:, dcols], off,
                   primary_field=table[:, pcols] if pcols else None, loop_angles=angles)

    def subset(self, rows):
        return type(self)(self.system, self.lineNumber[rows], self.fiducial[rows], self.x[rows], self.y[rows], self.z[rows],
                          self.elevation[rows], self.data[rows], self.offsets[rows],
                          primary_field=None if self.primary_field is None else self.primary_field[rows], loop_angles=self.loop_angles[rows])


class TempestData(TdemData):
    """Fixed-wing Tempest soundings (classes/data/dataset/TempestData.py): the same file layout with X and Z components, the
    per-sounding transmitter-receiver offsets and the primary-field columns PX / PZ.  The reference's Tempest data point works
    on TOTAL fields -- channel = secondary + the component's primary (Tempest_datapoint.py:106-123) -- with per-channel
    additive errors scaled by a multiplier per component (:161-176): ``total_field`` gives those channels."""

    def total_field(self, rows=None):
        rows = slice(None) if rows is None else rows
        d = self.data[rows]
        if self.primary_field is None:
            return d
        nc = self.primary_field.shape[1]
        return d + np.repeat(self.primary_field[rows], d.shape[1] // nc, axis=1)


class SurveyResult(dict):
    """Per-sounding results, rows in file order: line, fiducial, x, y, z, elevation, status (1 done, 2 failed to burn
    in), burned_in_iteration, iterations, acceptance, misfit, relative_error, additive_error, n_layers, best_* (highest
    posterior model: n_layers, edges, conductivity), layer_count_posterior [S, K + 1], interface_posterior
    [S, n_depth_bins], depth_bin_width, and -- when the hit map is kept -- mean_log10_conductivity [S, n_depth_bins] and
    the 5 / 50 / 95 % conductivity percentiles per depth cell."""

    def save(self, filename):
        from .hdf import save_npz
        save_npz(filename, self, threads=max(1, min(8, _usable_cores())))     # (.npz, deflate level 1, members side by side: hdf.save_npz)

    @classmethod
    def load(cls, filename):
        """A result written by ``save`` / ``save_lines`` (scalars come back as Python scalars)."""
        with np.load(filename, allow_pickle=False) as f:
            return cls({k: (f[k].item() if f[k].ndim == 0 else f[k]) for k in f.files})

    @classmethod
    def load_lines(cls, directory):
        """The per-line files of ``save_lines`` put back together, lines in ascending order (rows keep their order within a line)."""
        names = sorted((n for n in os.listdir(directory) if n.endswith(".npz")), key=lambda n: float(n[:-4]))
        parts = [cls.load(os.path.join(directory, n)) for n in names]
        S = [p["line"].size for p in parts]
        out = cls()
        for k, v in parts[0].items():
            per_row = isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == S[0]
            out[k] = np.concatenate([p[k] for p in parts], axis=0) if per_row else v
        return out

    def save_lines(self, directory):
        """One file per flight line, ``<line number>.npz`` (the reference writes ``<line number>.h5`` there)."""
        from .hdf import save_npz
        S = self["line"].size
        paths = []
        for ln in np.unique(self["line"]):
            m = self["line"] == ln
            part = {k: (v[m] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == S else v) for k, v in self.items()}
            paths.append(os.path.join(directory, "{}.npz".format(ln)))
            save_npz(paths[-1], part)
        return paths


def _hitmap_statistics(hitmap, log_mean_prior, half_width):
    """Mean and percentiles of log10 conductivity per depth cell from the hit map (the reference derives the same from
    its Histogram2D posterior): one kernel over the maps, geobipy_amd.hitmap."""
    from .hitmap import statistics
    return statistics(hitmap, log_mean_prior, half_width)


def _usable_cores():
    """Host cores this process may really use: the scheduler affinity capped by the cgroup CPU quota (a 16-CPU quota on a 256-thread
    host: more writer threads than that only contend)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def _container_kind(ds):
    return "tempest" if isinstance(ds, TempestData) else ("tdem" if isinstance(ds, TdemData) else "fdem")


class _LineWriter:
    """Fills one results container per flight line (geobipy_amd.hdf) from chunks of device rows; a line's container is compressed and
    written out by a writer thread, and dropped, as soon as its last sounding has arrived -- what is held is the open lines."""

    def __init__(self, directory, ds, o, dc, hitmap, container=None):
        from concurrent.futures import ThreadPoolExecutor
        from . import hdf
        os.makedirs(directory, exist_ok=True)
        self.hdf, self.directory, self.ds, self.o, self.hitmap = hdf, directory, ds, o, hitmap
        self.container = hdf.container_type(container)        # "hdf5" (<line>.h5) | "npz" (<line>.results.npz + .attrs.json)
        self.K, self.N, self.nd, self.nv = dc.K, dc.N, dc.n_depth_bins, dc.n_value_bins
        kind = self.kind = _container_kind(ds)
        td = kind != "fdem"
        n_primary = (ds.primary_field.shape[1] if getattr(ds, "primary_field", None) is not None else 0) if td else 0
        height = bool(getattr(dc, "solve_height", False))
        angles = tuple((m_[0], m_[5]) for m_ in (getattr(dc, "_moves", None) or ()))
        self.trace_every, self.trace_length = int(getattr(dc, "trace_every", 0) or 0), int(getattr(dc, "trace_length", 0) or 0)
        fkw = dict(hitmap=hitmap, n_rel=dc.n_rel_groups, n_add=dc.n_add_groups, time_domain=td, n_primary=n_primary, height=height, angles=angles,
                   trace_length=self.trace_length)
        ff, fi = hdf.device_row_fields(self.N, self.K, self.nd, self.nv, **fkw)
        # int32 rows come with dense hit-map columns (wi_dense: rows streamed from other ranks) or without (wi: the hit maps travel as
        # their non-zero entries); hm_tail = the columns after the hit map's
        self.wf, self.wi_dense = sum(w for _, w in ff), sum(w for _, w in fi)
        names_i = [n_ for n_, _ in fi]
        self.hm_tail = sum(w for _, w in fi[names_i.index("hitmap") + 1:]) if hitmap else 0
        self.wi = self.wi_dense - (self.nv * self.nd if hitmap else 0)
        self.line_col = [n_ for n_, _ in ff].index("line_number")
        self.fid_col = [n_ for n_, _ in ff].index("fiducial")
        self.wkw = dict(hitmap=hitmap, kind=kind, n_rel=dc.n_rel_groups, n_add=dc.n_add_groups, n_primary=n_primary,
                        loop_radius=ds.system[0].loopRadius() if td else 0.0,
                        channel_additive=o.get("initial_additive_error") if kind == "tempest" else None, height=height, angles=angles,
                        trace_length=self.trace_length)
        self.lines, self.paths = {}, []
        # finished lines are compressed and written by a few host threads (zlib releases the interpreter lock) while the next rows
        # arrive; at most 2 x workers lines wait for their turn, so the process still holds a bounded number of lines
        self.workers = max(1, min(16, _usable_cores() - 1))
        self.pool, self.pending = ThreadPoolExecutor(max_workers=self.workers), []

    def _close(self, ln):
        root, fid, path, _ = self.lines.pop(ln)
        if isinstance(root, self.hdf.NpzGroup):
            while len(self.pending) >= 2 * self.workers:
                self.pending.pop(0).result()
            self.pending.append(self.pool.submit(root.save, path))   # <line>.h5, or <line>.results.npz (+ <line>.results.attrs.json)
            self.paths.append(path if root.container == "hdf5" else path + ".npz")
        else:
            root.close()
            self.paths.append(path)

    def add(self, f, i, csr=None):
        """Rows of hdf.device_row_fields (numpy [m, wf] float64, [m, wi] int32) of any lines; ``csr`` = (ptr, index, value): the
        rows' hit maps in run-length form (hdf._Dataset.write_run_rows; the int32 rows then carry no hit-map columns)."""
        hdf, ds = self.hdf, self.ds
        if csr is None and self.hitmap:               # dense hit-map columns (rows streamed from other ranks): the same run-length form
            c0 = self.wi_dense - self.hm_tail - self.nv * self.nd
            hm = i[:, c0:c0 + self.nv * self.nd]
            edge = np.ones(hm.shape, dtype=bool)     # run starts: a row's first cell and every change of value
            edge[:, 1:] = hm[:, 1:] != hm[:, :-1]
            r_, j_ = np.nonzero(edge)
            csr = (np.r_[0, np.cumsum(np.bincount(r_, minlength=hm.shape[0]))], j_.astype(np.int32), hm[r_, j_])
            i = np.concatenate([i[:, :c0], i[:, c0 + self.nv * self.nd:]], axis=1)
        for ln in np.unique(f[:, self.line_col]):
            if ln not in self.lines:
                fid = np.sort(ds.fiducial[ds.lineNumber == ln])
                path = hdf.results_path(self.directory, ln, container=self.container)     # <line>.h5, or <line>.results(.npz) for the stand-in
                root = hdf.open_results(path, container=self.container)
                hdf.create_inference1d(root, hdf.LineSpec(ds.system, self.N, self.o, n_value_bins=self.nv, kind=self.kind,
                                                          trace_every=max(1, self.trace_every)), add_axis=fid)
                self.lines[ln] = [root, fid, path, 0]
            root, fid, _, _ = self.lines[ln]
            m = f[:, self.line_col] == ln
            rows_ = np.flatnonzero(m)
            run = rows_[-1] - rows_[0] + 1 == rows_.size       # the line's rows are one run of the chunk (the usual case): views, no copies
            fm, im = (f[rows_[0]:rows_[-1] + 1], i[rows_[0]:rows_[-1] + 1]) if run else (f[m], i[m])
            sub = csr
            if csr is not None and rows_.size != f.shape[0]:       # the line's rows of the chunk's run-length block
                ptr_, ind_, val_ = csr
                if run:
                    a_, b_ = int(ptr_[rows_[0]]), int(ptr_[rows_[-1] + 1])
                    sub = (ptr_[rows_[0]:rows_[-1] + 2] - a_, ind_[a_:b_], val_[a_:b_])
                else:
                    take = np.concatenate([np.arange(ptr_[q_], ptr_[q_ + 1]) for q_ in rows_])
                    sub = (np.r_[0, np.cumsum(np.diff(ptr_)[rows_])], ind_[take], val_[take])
            hdf.write_device_rows(root, np.searchsorted(fid, fm[:, self.fid_col]), fm, im, self.N, self.K, self.nd, self.nv, self.o,
                                  hitmap_csr=sub, **self.wkw)
            self.lines[ln][3] += int(rows_.size)
            if self.lines[ln][3] >= fid.size:
                self._close(ln)

    def add_block(self, block):
        """One finished block of this process: payload() -- (rows, float64 rows, int32 rows) on the host with dense hit-map columns,
        taken 64 rows at a time, or (rows, float64 rows, int32 rows, (ptr, index, value)) with the hit maps as their non-zero
        entries, taken whole (the rows of a line are then one slice of each array)."""
        f_b, i_b = block[1], block[2]
        csr = block[3] if len(block) > 3 else None
        f_np, i_np = f_b.numpy(), i_b.numpy()      # (host tensors: views; nothing collective here -- ranks that own whole lines call
        assert f_b.shape[1] == self.wf             #  this as often as they have blocks)
        if csr is not None:
            assert i_b.shape[1] == self.wi
            if f_np.shape[0]:
                self.add(f_np, i_np, csr)
            return
        assert i_b.shape[1] == self.wi_dense
        for a in range(0, f_np.shape[0], 64):
            self.add(f_np[a:a + 64], i_np[a:a + 64])

    def finish(self):
        for ln in list(self.lines):
            self._close(ln)
        for job in self.pending:
            job.result()                           # (re-raises what a writer thread raised)
        self.pool.shutdown()
        return self.paths


def _write_line_containers(directory, ds, o, dc, shipped, hitmap, rank, container=None):
    """Several ranks: rank 0 receives every rank's rows chunk by chunk (ONE exchange: every rank enters it once, whatever number of
    blocks the dynamic schedule gave it) and fills the line containers (_LineWriter).  (One process writes its blocks as they
    finish: infer.)"""
    import torch
    from .distributed import stream_rows_to_root
    w = _LineWriter(directory, ds, o, dc, hitmap, container)
    cat = lambda j, wd, dt: torch.cat([s_[j] for s_ in shipped]) if shipped else torch.zeros((0, wd) if wd else (0,), dtype=dt)
    rows_t, f_t, i_t = cat(0, 0, torch.int64), cat(1, w.wf, torch.float64), cat(2, w.wi_dense, torch.int32)
    assert f_t.shape[1] == w.wf and i_t.shape[1] == w.wi_dense
    if dc.device.type != "cpu" and torch.distributed.is_initialized() and torch.distributed.get_backend() != "gloo":
        rows_t, f_t, i_t = rows_t.to(dc.device), f_t.to(dc.device), i_t.to(dc.device)     # RCCL sends device memory, 64 rows at a time
    for _, (f, i) in stream_rows_to_root(rows_t, [f_t, i_t], chunk_rows=64):
        w.add(f, i)
    return w.finish()


def select_soundings(ds, index=None, fiducial=None, line_number=None):
    """Row indices chosen by the reference's --index / --fiducial / --line switches (Inference3D.infer_serial :458-500:
    one data point by position, or by fiducial on a line); all rows when none is given."""
    if index is not None:
        assert 0 <= index < ds.nPoints, ValueError("index {} outside the {} data points".format(index, ds.nPoints))
        return np.array([index])
    if fiducial is not None:
        assert line_number is not None, ValueError("--fiducial needs --line")
        hit = np.nonzero((ds.fiducial == fiducial) & (ds.lineNumber == line_number))[0]
        assert hit.size > 0, ValueError("fiducial {} not found on line {}".format(fiducial, line_number))
        return hit[:1]
    if line_number is not None:
        hit = np.nonzero(ds.lineNumber == line_number)[0]
        assert hit.size > 0, ValueError("line {} not found".format(line_number))
        return hit
    return np.arange(ds.nPoints)


BLOCK_PAYLOAD_BUDGET = 8 << 30      # bytes of traces + hit maps one device block may hold by default (infer's ``chunk``)


def infer(options, output=None, seed=None, device=None, hitmap=True, burn_in_min_iterations=5000, check_every=1000,
          exact_jacobian=False, data=None, index=None, fiducial=None, line_number=None, hankel_eps=None, schedule="static",
          chunk=None, results_directory=None, timings=None, traces=1, container=None, **overrides):
    """Invert every sounding of the options file's data set.  One process per GPU: call from every rank of an initialised
    ``torch.distributed`` group to shard the soundings (``distributed.shard``); rank 0 returns the SurveyResult of the
    whole survey (and writes ``output`` if given), the other ranks return None.

    ``chunk``: soundings per block on the device (default 16 384 for "static" and "lines", fewer when full-length traces and hit maps of
    that many soundings would exceed BLOCK_PAYLOAD_BUDGET on the device; see "dynamic").
    ``schedule``: "auto" (the command line's default) -- "lines" on more than one rank when containers are written, the data file allows
    it and whole lines balance over the ranks (most loaded rank <= 1.2 x the mean), else "static";
    "lines" -- whole flight lines per rank, longest first to the least loaded rank (``distributed.assign_lines``): every
    rank writes the results containers of its own lines and no posterior row travels (the choice for many GPUs with containers);
    "static" -- each rank inverts one contiguous block (``distributed.shard``); "dynamic" -- the ranks draw chunks
    of ``chunk`` soundings (default: a 16th of a rank's static share, at least 256) from a shared counter until none are left
    (``distributed.ChunkQueue``; the reference's master / worker loop, Inference3D.py:518-635, without a master), which evens
    out the different numbers of iterations soundings need.  Chains are keyed by the sounding's row in the data file, so the
    results do not depend on the schedule.
    ``results_directory``: also write the reference's per-line results containers there (``<line number>.h5``: the layout of
    Inference2D.createHdf / Inference1D.writeHdf, ``geobipy_amd.hdf`` -- real HDF5 files, written by h5py when it is installed and through
    the HDF5 C library otherwise (``geobipy_amd.h5lite``); ``<line number>.results.npz`` with the same dataset paths where neither exists.
    ``container`` = "hdf5" | "npz" | None (the environment's GBP_CONTAINER, else whichever can be written; ``hdf.container_type()`` says
    which) -- every sounding's posteriors (layer count, interface depth, error levels, conductivity-depth hit
    map), best model and its predicted data.  The rows travel to rank 0 in bounded chunks (``distributed.stream_rows_to_root``).
    ``traces``: per-iteration misfit / acceptance traces for the containers' ``phids`` / ``acceptance_rate``, kept on the device at a stride --
    1 (default): the reference's arrays in full, 2 n_markov_chains columns, the shapes ``Inference1D.createHdf`` writes and its readers
    index by iteration; an int > 1 or "auto" (opt-in: the smallest stride with at most 4 096 entries per sounding): every stride-th
    entry side by side, ceil(2 n_markov_chains / stride) columns with the stride as the datasets' attribute ``trace_every`` -- NOT the
    reference's shapes; None: none.
    ``timings``: a dict that receives the wall time by phase (the device is synchronised at the phase borders then; bench.py).
    ``index`` / ``fiducial`` + ``line_number`` / ``line_number``: the reference's single-point and single-line switches.
    ``exact_jacobian``: use the true derivative of the forward model in the proposals instead of the reference's
    expression.  ``hankel_eps``: accuracy-budgeted window of the Hankel filter abscissae.  Frequency domain: ppm, per
    sounding, default 1e-10 (``DeviceChains(hankel_eps_ppm=...)``; 0 = all abscissae).  Time domain: relative to the
    inductive-limit value of every nodal sum, per sounding, default 1e-12 (``TdemDeviceChains(hankel_eps=...)``; 0 = all)."""
    import torch
    import torch.distributed as dist
    from .distributed import shard
    from .rjmcmc_gpu import DeviceChains

    o = read_options(options, **overrides) if isinstance(options, str) else dict(options)
    tempest = o["data_type"] in ("TempestData", "Tempest_datapoint")
    time_domain = tempest or o["data_type"] in ("TdemData", "TdemDataPoint")
    if not time_domain and o["data_type"] not in ("FdemData", "FdemDataPoint"):
        raise NotImplementedError("the device sampler handles FdemData, TdemData and TempestData; {} is not supported".format(o["data_type"]))
    geometry_keys = [k_ for k_ in o if (k_.startswith("solve_transmitter_") or k_.startswith("solve_receiver_")) and o[k_]]
    if geometry_keys and not time_domain:
        raise NotImplementedError(geometry_keys[0] + ": frequency-domain data points have no loop pair to sample")
    # (time-domain data: the loops' attitude angles are sampled on the device, gbp_td_moves; position moves raise in TdemDeviceChains)
    if o.get("solve_calibration"):
        raise NotImplementedError("solve_calibration is not supported by the device sampler")
    if o.get("ignore_likelihood") and time_domain:
        raise NotImplementedError("ignore_likelihood (prior-only sampling) on time-domain data is not supported by the device sampler")
    # (frequency-domain data: DeviceChains(ignore_likelihood=True) -- the prior alone, Inference1D.py:394, 519, 551, 596)
    # solve_height: the reference's datapoint only moves its height for the keys solve_z / maximum_z_change /
    # z_proposal_variance (pointcloud/Point.py:949-983), which its options files never set -- with the files as shipped the height
    # stays fixed there too.  An options file that DOES carry solve_z = True gets the move (frequency-domain data; DeviceChains).
    if time_domain and o.get("solve_z"):
        raise NotImplementedError("solve_z on time-domain data: the reference's forward takes the TRANSMITTER's z (system/Loop_pair.py:70), "
                                  "which the data point's z move never touches -- the key that would matter is solve_transmitter_z, and the "
                                  "geometry of the loop pair is not sampled")
    if data is not None:
        ds = data
    elif time_domain:
        ds = (TempestData if tempest else TdemData).read_csv(o["data_filename"], o["system_filename"])
    else:
        ds = FdemData.read_csv(o["data_filename"], o["system_filename"])
    rows = select_soundings(ds, index, fiducial, line_number)
    if rows.size != ds.nPoints:
        ds = ds.subset(rows)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    start, n = shard(ds.nPoints, rank, world)
    sl = slice(start, start + n)
    seed = o.get("seed", 0) if seed is None else seed
    keys = ("ignore_likelihood", "n_markov_chains", "solve_gradient", "solve_parameter", "solve_relative_error", "solve_additive_error", "maximum_number_of_layers",
            "minimum_depth", "maximum_depth", "minimum_thickness", "initial_relative_error", "minimum_relative_error",
            "maximum_relative_error", "initial_additive_error", "minimum_additive_error", "maximum_additive_error",
            "relative_error_proposal_variance", "additive_error_proposal_variance", "probability_of_birth",
            "probability_of_death", "probability_of_perturb", "probability_of_no_change", "factor",
            "gradient_standard_deviation", "covariance_scaling", "parameter_limits", "update_plot_every", "reset_limit",
            "solve_z", "maximum_z_change", "z_proposal_variance")
    if time_domain:
        from .tdem_geometry import LOOP_PAIR_SCALARS
        keys = keys + tuple(k_ for _, stem, _ in LOOP_PAIR_SCALARS for k_ in ("solve_" + stem, "maximum_" + stem + "_change", stem + "_proposal_variance"))
    # chains are keyed by the sounding's row in the data file, so a sounding inverted alone walks the chain it walks in the
    # full survey
    assert rows.size == 1 or np.all(np.diff(rows) == 1), "selected soundings must be contiguous rows"
    common = dict(seed=int(seed) % (1 << 64), device=device, hitmap=hitmap, first_chain=int(rows[0]) + start, reference_schedule=True,
                  burn_in_min_iterations=burn_in_min_iterations, **{k: o[k] for k in keys if o.get(k) is not None})
    # per-iteration traces for the containers' `phids` / `acceptance_rate` (Inference1D.data_misfit_v / acceptance_v): kept on the
    # device at a stride -- "auto": the smallest stride with at most 4 096 entries per sounding (32 + 4 KB per sounding beside a
    # 440 KB hit map; the reference's full arrays are 2 n_markov_chains x 9 bytes = 1.8 MB at its default 100 000); an int: that
    # stride (1 = the reference's arrays in full); None / 0: no traces (the two datasets stay at their fill values)
    if results_directory is not None and traces:
        n_mc2 = 2 * int(o["n_markov_chains"])
        common.update(trace_every=max(1, -(-n_mc2 // 4096)) if traces == "auto" else int(traces))
    # Default block size: 16 384 soundings, less when a sounding's posterior payload on the device is large -- full-length traces at the
    # reference's default n_markov_chains = 100 000 are 1.8 MB per sounding (29.5 GB for 16 384, plus their host copies): the default
    # block keeps traces + hit maps under BLOCK_PAYLOAD_BUDGET.  Chains are keyed by row, so the block size never changes a result.
    def default_block(limit=16384):
        per = 0
        if common.get("trace_every"):
            per += -(-2 * int(o["n_markov_chains"]) // int(common["trace_every"])) * 9        # misfit f64 + acceptance u8 per kept entry
        if hitmap and results_directory is not None:
            per += 440 * 1024                                                                 # (the hit map's usual size; exact: DeviceChains)
        return limit if per == 0 else int(max(256, min(limit, BLOCK_PAYLOAD_BUDGET // per)))
    if time_domain and hankel_eps is not None:
        common.update(hankel_eps=float(hankel_eps))
    elif not time_domain and hankel_eps is not None:
        common.update(hankel_eps_ppm=float(hankel_eps))
    f64 = lambda x: x.to(torch.float64)
    col = lambda x: f64(x)[:, None]
    # wall time by phase (device-synchronised at the phase borders only when a caller asks for it with timings={}: bench.py's
    # ``survey`` object; a normal run never synchronises for this)
    import time as _time

    class _Phase:
        def __init__(self, name):
            self.name = name
        def __enter__(self):
            if timings is not None:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                self.t0 = _time.perf_counter()
        def __exit__(self, *a):
            if timings is not None:
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                timings[self.name] = timings.get(self.name, 0.0) + _time.perf_counter() - self.t0

    def run_block(idx, offset=None):
        """Chains of the soundings ``idx`` (rows of ds, ascending) to completion -> (sampler, [(name, [len(idx), w])])."""
        kw = dict(common)
        if idx.size != n or idx[0] != start:        # a selection of the shard: key every chain by its own row of the data file
            kw.pop("first_chain")
            kw["chain_id"] = int(rows[0]) + idx
        if time_domain:
            from .tdem import TdemDeviceChains
            if isinstance(ds, TempestData):
                # Tempest_datapoint (data/datapoint/Tempest_datapoint.py:106-123, 161-176): the channels hold primary + secondary
                # field, the options file's additive errors are per channel and the sampled level is their multiplier per component
                nc = ds.system[0].n_components
                kw.update(channel_additive=np.asarray(o["initial_additive_error"], dtype=np.float64), initial_additive_error=[1.0] * nc,
                          primary_field=ds.primary_field[idx] if ds.primary_field is not None else None)
            dc = TdemDeviceChains(ds.system, ds.z[idx], ds.total_field(idx) if isinstance(ds, TempestData) else ds.data[idx], offset,
                                  attitude=ds.attitude[idx] if idx.size else None, **kw)
        else:
            with _Phase("upload_and_initialise"):
                dc = DeviceChains(ds.system, ds.z[idx], ds.data[idx], exact_jacobian=exact_jacobian, **kw)
        with _Phase("chains"):
            dc.infer(check_every=check_every)
        t = dc.t
        named = [("status", col(t["status"])), ("burned_in_iteration", col(t["burned_in_iteration"])), ("n_accepted", col(t["n_accepted"])),
                 ("misfit", col(t["misfit"])), ("relative_error", t["rel"]), ("additive_error", t["add"]), ("n_layers", col(t["k"])),
                 ("best_n_layers", col(t["best_k"])), ("best_posterior", col(t["best_posterior"])), ("best_edges", t["best_edges"]),
                 ("best_conductivity", t["best_sigma"]), ("layer_count_posterior", f64(t["k_hist"])),
                 ("interface_posterior", f64(t["edge_hist"])), ("relative_error_posterior", f64(t["rel_hist"]).flatten(1)),
                 ("additive_error_posterior", f64(t["add_hist"]).flatten(1))]
        if getattr(dc, "_moves", None):            # sampled attitude angles (the loops' own convention): final, highest-posterior, posterior
            cur, best = dc.sampled_angles("geom"), dc.sampled_angles("best_geom")
            for q, m_ in enumerate(dc._moves):
                named += [(m_[0], col(cur[m_[0]])), ("best_" + m_[0], col(best[m_[0]])), (m_[0] + "_posterior", f64(t["geom_hist"][:, q, :m_[5]]))]
        if getattr(dc, "solve_height", False):     # the sampled height: final and highest-posterior values, posterior on the prior's 99 cells
            named += [("height", col(t["height"])), ("best_height", col(t["best_height"])), ("height_posterior", f64(t["height_hist"]))]
        if hitmap:
            with _Phase("hitmap_statistics"):
                mean, pct = _hitmap_statistics(dc.hitmap, t["log_mean_prior"], dc.value_half_width)     # (attribute access settles dwell times)
            named += [("mean_log10_conductivity", mean)] + [("log10_conductivity_" + q, p) for q, p in zip(("p05", "p50", "p95"), pct)]
        return dc, named

    state = dict(iterations=0, dc=None, named=None)
    shipped = []                                   # per block: (rows, float64 block, int32 block) of hdf.device_row_fields, on the HOST

    def payload(dc, idx, sparse=False):
        """The rows of hdf.device_row_fields for a finished block, moved to host memory at once (the hit maps are 440 KB per
        sounding: what stays on the GPU is the running block, not every block a rank has finished).  ``sparse``: the hit maps leave
        the device in run-length form (per row: the flat positions value_bin * n_depth + depth cell at which the count changes, and
        the counts; hdf._Dataset.write_run_rows) instead of dense int32 columns -- depth is the fast axis and a layer fills a run of
        cells with one count: a few thousand runs against 110 000 cells -- for a process that fills its own containers."""
        from .rjmcmc_gpu import layer_widths
        t, dev = dc.t, dc.device
        n_mc = int(o["n_markov_chains"])
        none = t["best_k"] < 1                      # (a chain that never recorded a best model: its current one)
        bk = torch.where(none, t["k"], t["best_k"])
        be = torch.where(none[:, None], t["edges"], t["best_edges"])
        bs = torch.where(none[:, None], t["sigma"], t["best_sigma"])
        # the error levels of the highest-posterior state, like Inference1D.writeHdf's best data point (:1076-1088)
        brel = torch.where(none[:, None], t["rel"], t["best_rel"]).contiguous()
        badd = torch.where(none[:, None], t["add"], t["best_add"]).contiguous()
        observed = dc.observed                     # (the measured data: t["data"] unless the chains sampled the prior alone)
        pred = torch.empty_like(observed)
        chi2, logl = torch.empty_like(t["misfit"]), torch.empty_like(t["misfit"])
        # sampled attitude angles: the best data point's OWN geometry -- the prediction and the predicted primary field of the
        # highest-posterior angles, not of the chain's last state / the measured geometry (Inference1D.writeHdf :1076-1088 writes
        # the best data point: predicted_secondary_field = predictedData - predicted_primary_field there)
        best_mix = {}
        best_primary = None
        eval_height = t["best_height"] if t.get("best_height") is not None else t["height"]
        if getattr(dc, "_moves", None):
            bw, boff, best_primary = dc.mix_for_geometry(torch.where(none[:, None], t["geom"], t["best_geom"]))
            best_mix = dict(weights=bw, offset=boff)
            extra = dc.geometry_rows_extra()       # sampled positions: the best state's distance scale and effective height
            if extra is not None:
                best_mix["scale"], eval_height = extra["scale"], extra["height"]
        with torch.cuda.device(dev):                # one batched forward at the best models, through the sampler's own entry
            dc._eval_loglike(bk.contiguous(), bs.contiguous(), layer_widths(be, bk.to(torch.int64)).contiguous(),
                             eval_height, observed, brel, badd, pred, chi2, logl, **best_mix)
        host = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)[idx], device=dev).reshape(idx.size, -1)
        cols_f = [host(ds.x), host(ds.y), host(ds.z), host(ds.elevation), host(ds.lineNumber), host(ds.fiducial), observed, pred,
                  brel, badd, t["log_mean_prior"][:, None], be, bs]
        if time_domain:
            n_pf = ds.primary_field.shape[1] if ds.primary_field is not None else 0
            cols_f += [dc.channel_std(observed, brel, badd), host(ds.offsets), host(ds.loop_angles)]
            if n_pf:
                cols_f += [host(ds.primary_field),
                           torch.as_tensor(dc.predicted_primary() if best_primary is None else best_primary, device=dev).reshape(idx.size, -1)]
        if getattr(dc, "solve_height", False):
            cols_f += [t["best_height"][:, None], t["height0"][:, None]]
        if getattr(dc, "_moves", None):
            bst, ctr = dc.sampled_angles("best_geom"), dc.sampled_angles("geom0")
            for m_ in dc._moves:
                cols_f += [bst[m_[0]][:, None], ctr[m_[0]][:, None]]
        if getattr(dc, "trace_every", 0):
            cols_f.append(t["trace_misfit"])
        f64_block = torch.cat(cols_f, dim=1).contiguous()
        st, bi = t["status"].to(torch.int32), t["burned_in_iteration"].to(torch.int32)
        ran = torch.where(st == 1, bi + n_mc + 1, torch.where(st == 2, torch.full_like(bi, n_mc), torch.full_like(bi, dc.iteration)))
        cols = [st[:, None], bi[:, None], ran[:, None], bk.to(torch.int32)[:, None], t["best_iteration"][:, None], t["k_hist"], t["edge_hist"], t["rel_hist"].flatten(1),
                t["add_hist"].flatten(1)]
        csr = None
        if hitmap and sparse:
            from .hitmap import runs                # run starts: a row's first cell and every change of value (csrc/gbp_hitmap.h)
            ptr, start_, val_ = runs(dc.hitmap)     # (attribute access settles the dwell times)
            csr = (ptr.cpu().numpy(), start_.cpu().numpy(), val_.cpu().numpy())
            del start_, val_
        elif hitmap:
            cols.append(dc.hitmap.flatten(1))       # (attribute access settles the dwell times)
        if getattr(dc, "solve_height", False):
            cols.append(t["height_hist"])
        for q_, m_ in enumerate(getattr(dc, "_moves", None) or ()):
            cols.append(t["geom_hist"][:, q_, :m_[5]])
        if getattr(dc, "trace_every", 0):
            cols.append(t["trace_accept"])
        to_host = lambda x: x.cpu()
        out = (torch.as_tensor(np.asarray(idx), dtype=torch.int64), to_host(f64_block),
               to_host(torch.cat([c_.to(torch.int32) for c_ in cols], dim=1).contiguous()))
        return out + (csr,) if sparse else out

    def fill_containers(background=False):
        """The finished block's rows -> host -> line containers.  ``background``: on a host thread with a device stream of its own, while
        the caller's thread runs the NEXT block's chains (the block's sampler stays alive until its rows have left the device; one block is
        in flight at a time, so the containers receive the blocks in order) -- 65 536 soundings: the rows of three of the four blocks no
        longer stand between two blocks' chains.  The phase clocks of a background fill are host wall time, overlapped with "chains"."""
        join_fill()
        if state.get("unfilled") is None:
            return
        dc_, idx_ = state.pop("unfilled")
        if state.get("writer") is None:
            state["writer"] = _LineWriter(results_directory, ds, o, dc_, hitmap, container)
        if idx_.size == 0:                          # (a rank that got no flight line: nothing to hand over)
            return
        if not background or dc_.device.type != "cuda":
            with _Phase("rows_to_host"):
                pl = payload(dc_, idx_, sparse=True)
            with _Phase("container_fill"):
                state["writer"].add_block(pl)
            return
        import threading
        side = state.get("fill_stream")
        if side is None:
            side = state["fill_stream"] = torch.cuda.Stream(device=dc_.device)
        side.wait_stream(torch.cuda.current_stream(dc_.device))      # (the block's chains have ended: infer() read their status flags)
        failed = state.setdefault("fill_failed", [])

        def work():
            try:
                t0_ = _time.perf_counter()
                with torch.cuda.device(dc_.device), torch.cuda.stream(side):
                    pl = payload(dc_, idx_, sparse=True)
                t1_ = _time.perf_counter()
                state["writer"].add_block(pl)
                if timings is not None:
                    timings["rows_to_host_overlapped"] = timings.get("rows_to_host_overlapped", 0.0) + t1_ - t0_
                    timings["container_fill_overlapped"] = timings.get("container_fill_overlapped", 0.0) + _time.perf_counter() - t1_
            except BaseException as e:               # (handed to the caller's thread by join_fill)
                failed.append(e)
        th = state["fill_thread"] = threading.Thread(target=work)
        th.start()

    def join_fill():
        th = state.pop("fill_thread", None)
        if th is not None:
            th.join()
        if state.get("fill_failed"):
            raise state["fill_failed"].pop(0)

    def process(first, count):
        """Result rows [count, width] of the soundings first .. first + count - 1 (count >= 0)."""
        span = np.arange(first, first + count)
        if time_domain:
            # the Hankel tables depend on the horizontal transmitter-receiver distance and dz: the block's handle holds one table
            # set per distinct pair and every chain runs with its own; azimuth and attitude are per-chain mixing weights
            # (TdemDeviceChains(offset=[n, 3], attitude=[n, 6])) -- one block whatever the geometry
            n_off = np.unique(np.c_[np.hypot(ds.offsets[span, 0], ds.offsets[span, 1]), ds.offsets[span, 2]], axis=0).shape[0] if count > 0 else 1
            if n_off > TdemData.MAX_OFFSET_SETS:
                raise NotImplementedError("{} distinct (horizontal distance, dz) receiver offsets in {} soundings: the device sampler holds one "
                                          "set of Hankel tables (~0.15 MB x (1 + altitude bins)) per pair -- bin the offsets (e.g. to 0.1 m) first".format(n_off, count))
            blocks = [(ds.offsets[span] if count > 0 else (0.0, 0.0, 0.0), span)]
        else:
            blocks = [(None, span)]
        out = None
        for off, idx in blocks:
            fill_containers(background=True)        # (the block before this one, if any: its rows leave while this block's chains run)
            dc, named = run_block(idx, off)
            if results_directory is not None and (world == 1 or schedule == "lines"):
                # one process: the block's rows go to the line containers and are dropped (host memory holds the open lines, not the
                # survey's hit maps) -- at the start of the next block, or, for the last one, once the summary file's thread is running
                state["unfilled"] = (dc, idx)
            elif results_directory is not None and idx.size:      # (a rank that drew no chunk ships nothing)
                shipped.append(payload(dc, idx))
            part = torch.cat([v for _, v in named], dim=1).contiguous()
            state.update(iterations=max(state["iterations"], dc.iteration), dc=dc, named=named)
            if len(blocks) == 1:
                return part
            if out is None:
                out = torch.empty((count, part.shape[1]), dtype=torch.float64, device=part.device)
            out[torch.as_tensor(idx - first, device=part.device)] = part
        return out

    assert schedule in ("auto", "static", "dynamic", "lines"), ValueError("schedule must be 'auto', 'static', 'dynamic' or 'lines'")
    if schedule == "auto":
        # more than one rank: whole lines per rank -- every rank writes its own containers and the job's only exchange is the gather of the
        # one-row summaries (all_gather_into_tensor; the posterior rows of "static" / "dynamic" travel point to point, which has run over
        # gloo only) -- whenever the data file allows it (every flight line one run of consecutive rows)
        change_ = np.flatnonzero(np.diff(ds.lineNumber) != 0) + 1
        firsts_ = np.r_[0, change_] if ds.nPoints else np.zeros(0, dtype=np.int64)
        lines_ok = bool(world > 1 and results_directory is not None and ds.nPoints and np.unique(ds.lineNumber[firsts_]).size == firsts_.size
                        and firsts_.size >= world)
        if lines_ok:
            # ... and only when whole lines balance: the most loaded rank within 1.2 x the mean (a survey with fewer lines than ranks,
            # or one dominant line, would leave GPUs idle where "static" uses all of them); without containers "lines" buys nothing
            from .distributed import assign_lines as _assign
            counts_ = np.diff(np.r_[firsts_, ds.nPoints])
            loads = [int(sum(counts_[i] for i in mine_)) for mine_ in _assign(counts_, world)]
            lines_ok = max(loads) <= 1.2 * ds.nPoints / world
        schedule = "lines" if lines_ok else "static"
    if schedule == "lines":
        # whole flight lines per rank: every rank fills and writes the results files of its own lines (as the reference's ranks write
        # their own rows, Inference3D.py:586-635), only the one-row summaries are gathered
        from .distributed import assign_lines, gather_rows
        change = np.flatnonzero(np.diff(ds.lineNumber) != 0) + 1
        firsts = np.r_[0, change] if ds.nPoints else np.zeros(0, dtype=np.int64)
        counts = np.diff(np.r_[firsts, ds.nPoints])
        if np.unique(ds.lineNumber[firsts]).size != firsts.size:
            raise ValueError("schedule='lines' needs every flight line in one run of consecutive rows of the data file (use 'static' or 'dynamic')")
        n = -1                                      # (every block is a selection: chains keyed by chain_id)
        size = int(chunk) if chunk else default_block()   # a long line goes through the device in pieces of this many soundings
        done_rows, done_vals = [], []
        for li in assign_lines(counts, world)[rank]:
            for first in range(int(firsts[li]), int(firsts[li] + counts[li]), size):
                count = min(size, int(firsts[li] + counts[li]) - first)
                done_vals.append(process(first, count))
                done_rows.append(torch.arange(first, first + count, dtype=torch.int64, device=done_vals[-1].device))
        if not done_vals:                           # more ranks than lines: an empty block fixes the row width and the device
            done_vals.append(process(0, 0))
            done_rows.append(torch.zeros(0, dtype=torch.int64, device=done_vals[-1].device))
        gathered = gather_rows(torch.cat(done_rows), torch.cat(done_vals), ds.nPoints)
    elif schedule == "static":
        # a rank's block goes through the device in pieces of `chunk` soundings (default 16 384): the posteriors of a piece (440 KB of hit
        # map per sounding) leave the GPU, and with one process the host, before the next piece runs -- 65 536 soundings: 19.6 s and 16 GB
        # of host memory in pieces against 24.7 s and 38 GB in one block (scripts/bench_survey.py); the chains are keyed by row either way
        piece = int(chunk) if chunk else default_block()
        if n > piece:
            local = torch.cat([process(first, min(piece, start + n - first)) for first in range(start, start + n, piece)])
        else:
            local = process(start, n)
        if world > 1:                               # the one exchange of the job: per-sounding result rows to rank 0
            from .distributed import SummaryGather
            g = SummaryGather(ds.nPoints, local.shape[1], local.device)
            gathered = g.finish(g.launch(*[local[:, i] for i in range(local.shape[1])]))
        else:
            gathered = local
    else:
        from .distributed import ChunkQueue, gather_rows
        n = -1                                      # (every block is a selection: chains keyed by chain_id)
        size = int(chunk) if chunk else default_block(max(256, -(-ds.nPoints // (16 * world))))
        done_rows, done_vals = [], []
        for first, count in ChunkQueue(ds.nPoints, size):
            done_vals.append(process(first, count))
            done_rows.append(torch.arange(first, first + count, dtype=torch.int64, device=done_vals[-1].device))
        if not done_vals:                           # this rank got no chunk: an empty block fixes the row width and the device
            done_vals.append(process(0, 0))
            done_rows.append(torch.zeros(0, dtype=torch.int64, device=done_vals[-1].device))
        gathered = gather_rows(torch.cat(done_rows), torch.cat(done_vals), ds.nPoints)
    iterations_run, dc, named = state["iterations"], state["dc"], state["named"]
    if world > 1:                                   # an unfinished chain's count is the longest run of any rank
        it = torch.tensor([iterations_run], dtype=torch.int64, device=dc.device)
        dist.all_reduce(it, op=dist.ReduceOp.MAX)
        iterations_run = int(it)
    def finish_containers():
        if results_directory is not None and (world == 1 or schedule == "lines"):
            fill_containers()
            if state.get("writer") is None:         # (no sounding at all: the empty set of containers)
                state["writer"] = _LineWriter(results_directory, ds, o, dc, hitmap, container)
            with _Phase("compress_and_write_tail"):
                state["writer"].finish()
        elif results_directory is not None:
            _write_line_containers(results_directory, ds, o, dc, shipped, hitmap, rank, container)
    if rank != 0:
        finish_containers()
        return None
    with _Phase("summaries_to_host"):
        r = gathered.cpu().numpy()
    # the result is put together and the summary file compressed on a thread of its own while this one fills the line containers and
    # their writer threads finish (numpy's conversions and zlib release the interpreter lock)
    import threading
    failed, made = [], []

    def assemble_and_save():
        try:
            res = SurveyResult(line=ds.lineNumber, fiducial=ds.fiducial, x=ds.x, y=ds.y, z=ds.z, elevation=ds.elevation,
                               depth_bin_width=np.float64(dc.depth_bin_width))
            c0 = 0
            ints = ("status", "burned_in_iteration", "n_layers", "best_n_layers", "layer_count_posterior", "interface_posterior",
                    "relative_error_posterior", "additive_error_posterior", "height_posterior") + tuple(
                n_ + "_posterior" for n_ in ("dx", "dy", "dz", "tx_z", "tx_pitch", "tx_roll", "tx_yaw", "rx_pitch", "rx_roll", "rx_yaw"))
            for name, v in named:
                w = v.shape[1]
                block = r[:, c0:c0 + w]
                c0 += w
                if name in ints:
                    block = block.astype(np.int64)
                res[name] = block[:, 0] if w == 1 else block
            for name, G in (("relative_error_posterior", dc.n_rel_groups), ("additive_error_posterior", dc.n_add_groups)):
                if G > 1:                                   # [S, groups, cells]; ne cells, uniform in log10 between the prior bounds
                    res[name] = res[name].reshape(-1, G, dc.n_error_bins)
            n_mc = int(o["n_markov_chains"])             # iterations each chain ran before it froze (infer :641-688)
            ran = np.where(res["status"] == 1, res["burned_in_iteration"] + n_mc + 1, np.where(res["status"] == 2, n_mc, iterations_run))
            res["iterations"] = ran.astype(np.int64)
            res["acceptance"] = res.pop("n_accepted") / np.maximum(1, ran)
            for k_ in ("status", "burned_in_iteration", "n_layers", "best_n_layers"):
                res[k_] = res[k_].astype(np.int32)
            made.append(res)
            if output is not None:
                res.save(output)
        except BaseException as e:                       # (handed to the caller's thread below)
            failed.append(e)
    saver = threading.Thread(target=assemble_and_save)
    saver.start()
    try:
        finish_containers()
    finally:
        with _Phase("summary_file_tail"):
            saver.join()
    if failed:
        raise failed[0]
    return made[0]
