#!/usr/bin/env python
"""Reference values of the hit-map summary statistics (build container only; needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_hitmap_stats.py   ->  tests/golden/hitmap_stats.npz

The IMPORTED reference initialises an ``Inference1D`` on one Resolve sounding with its own options file (the conductivity-depth
posterior ``model.values.posterior`` is then the ``Histogram`` over the ``RectilinearMesh2D`` that ``Model.set_posteriors``
builds, model/Model.py:665-684: 250 value cells relative to the prior mean, the depth grid of the interface posterior).  A few
hit maps -- layered posteriors like the chains leave, a single cell, an empty column, counts that put the cumulative share
EXACTLY on a percentile -- are assigned to its counts and the reference's own ``Histogram.mean(axis)`` and
``Histogram.percentile(percent, axis)`` (statistics/Histogram.py:262-284, 369-401 -> mesh/Mesh.py:80-113, 173-215) are recorded,
together with the grid.  The fixture holds data only: counts, grid edges, the reference's outputs.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402


def main():
    import numpy as np
    import_reference()
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    options = up.user_parameters.read(opt_file, data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    options["n_markov_chains"] = 100
    options["save_hdf5"] = False
    options["interactive_plot"] = True
    options["update_plot_every"] = 100000
    data = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"])
    dp = data.datapoint(30)
    inf = Inference1D(prng=get_prng(seed=options["seed"] if "seed" in options else 1), **options)
    inf.initialize(dp)
    h = inf.model.values.posterior
    mesh = h.mesh
    x_edges = np.asarray(mesh.x.edges, dtype=np.float64)          # value axis (relative to the prior mean, log10)
    y_edges = np.asarray(mesh.y.edges, dtype=np.float64)          # depth
    rel_to = float(np.asarray(mesh.x.relative_to).ravel()[0])
    shape = tuple(h.counts.shape)
    nx, ny = mesh.x.nCells.item(), mesh.y.nCells.item()
    value_axis = shape.index(nx) if nx != ny else 0
    print("counts shape", shape, "value cells", nx, "depth cells", ny, "value axis", value_axis, "relative_to", rel_to,
          "x log", mesh.x.log, "edges", x_edges[:2], x_edges[-2:])

    rng = np.random.default_rng(11)
    maps = []
    for kind in range(6):
        c = np.zeros((nx, ny), dtype=np.int32)                   # [value, depth] -- the device's layout
        if kind in (0, 1, 2):
            for _ in range((12, 60, 200)[kind]):
                v = rng.integers(0, nx)
                lo, hi = np.sort(rng.integers(0, ny, 2))
                c[v, lo:hi + 1] += rng.integers(1, 900)
        elif kind == 3:                                          # one cell per depth
            c[rng.integers(0, nx, ny), np.arange(ny)] = 7
        elif kind == 4:                                          # exact ties: 20 samples, the 5 % / 50 % / 95 % points fall ON a cumulative share
            for z in range(ny):
                vs = np.sort(rng.choice(nx, 4, replace=False))
                c[vs, z] = (1, 9, 9, 1)
        elif kind == 5:                                          # empty columns among full ones
            for z in range(0, ny, 3):
                c[rng.integers(0, nx, 5), z] += rng.integers(1, 50, 5).astype(np.int32)
        maps.append(c)
    out = dict(x_edges=x_edges, y_edges=y_edges, relative_to=rel_to, value_axis=value_axis, counts=np.stack(maps))
    means, pcts = [], []
    for c in maps:
        h.values = (c if value_axis == 0 else c.T).astype(np.int32)
        m = np.asarray(h.mean(axis=value_axis).values, dtype=np.float64)
        p = [np.asarray(h.percentile(percent=q, axis=value_axis).values, dtype=np.float64) for q in (5.0, 50.0, 95.0)]
        means.append(m)
        pcts.append(np.stack(p))
    out["ref_mean"] = np.stack(means)                            # in S/m (the reference returns 10 ** (mean of log10 + relative_to))
    out["ref_percentiles"] = np.stack(pcts)                      # [map, 3, depth]
    np.savez_compressed(os.path.join(HERE, "hitmap_stats.npz"), **out)
    print("wrote hitmap_stats.npz", {k: np.asarray(v).shape for k, v in out.items()})
    print("mean[0][:4]", out["ref_mean"][0][:4], "pct[0][:, :4]", out["ref_percentiles"][0][:, :4])


if __name__ == "__main__":
    main()
