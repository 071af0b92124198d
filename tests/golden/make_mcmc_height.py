#!/usr/bin/env python
"""The reference's height move (build container only) -> tests/golden/mcmc_height.npz.

No options file of the reference sets the keys its data point reads for the height move (``solve_z``, ``maximum_z_change``,
``z_proposal_variance``: pointcloud/Point.py:949-983; the ``solve_height`` keys of the options files are never read), so the move is
switched on here the way a user would: by adding those keys to the options.  Runs the reference's own Inference1D
(inversion/Inference1D.py:537-631 with Point.perturb :614-621 inside DataPoint.perturb :531-573) on resolve_glacial.csv row 30
from its seed and records per iteration: decision, layer count, misfit, height, relative and additive error -- and at the end the
height posterior (bin edges as the reference stores them, counts) and the prior / proposal parameters.
Only inputs / outputs are stored (data), none of the reference's code.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, SUP, import_reference   # noqa: E402

N_IT = 600


def main():
    import numpy as np
    import_reference()
    from geobipy import FdemData, Inference1D, get_prng
    from geobipy.src.inversion import user_parameters as up

    opt_file = REF + "/documentation_source/source/supplementary/options_files/resolve_options"
    options = up.user_parameters.read(opt_file, data_directory=SUP)
    options["system_filename"] = SUP + "/resolve.stm"
    options.update(n_markov_chains=N_IT, save_hdf5=True, interactive_plot=False, update_plot_every=5000,
                   solve_z=True, maximum_z_change=1.0, z_proposal_variance=0.01)
    dp = FdemData.read_csv(SUP + "/resolve_glacial.csv", system=options["system_filename"]).datapoint(30)
    inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
    inf.initialize(dp)
    d = inf.datapoint
    st = inf.prng.bit_generator.state                                    # generator state at the start of iteration 0
    sv, inc, m64 = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
    rng_state = np.array([sv >> 64, sv & m64, inc >> 64, inc & m64, st["has_uint32"], st["uinteger"]], dtype=np.uint64)
    z0 = float(np.squeeze(d.z))
    out = dict(rng_state=rng_state, z0=z0, halfspace=float(inf.model.values[0]), data=np.asarray(d.data, dtype=np.float64).copy(),
               prior0=float(inf.prior), like0=float(inf.likelihood), misfit0=float(inf.data_misfit),
               z_prior=np.array([float(d.z.prior.min), float(d.z.prior.max)]),
               z_proposal_variance=float(np.squeeze(d.z.proposal.variance)))
    rows = []
    for _ in range(N_IT):
        inf.accept_reject()
        inf.update()
        d = inf.datapoint
        rows.append([float(bool(inf.accepted)), float(inf.model.nCells.item()), float(inf.data_misfit), float(np.squeeze(d.z)),
                     float(np.squeeze(d.relative_error)), float(np.squeeze(d.additive_error)), float(inf.prior), float(inf.likelihood)])
    rows = np.array(rows)
    post = inf.datapoint.z.posterior
    out.update(rows=rows, z_hist_edges=np.asarray(post.mesh.edges, dtype=np.float64), z_hist_edges_absolute=np.asarray(post.mesh.edges_absolute, dtype=np.float64),
               z_hist_counts=np.asarray(post.values, dtype=np.int64), z_hist_relative_to=float(np.squeeze(post.mesh.relative_to)),
               burned_in=int(bool(inf.burned_in)), iteration=int(inf.iteration))
    np.savez_compressed(HERE + "/mcmc_height.npz", **out)
    print("accepted", int(rows[:, 0].sum()), "final k", rows[-1, 1], "misfit", rows[-1, 2], "z from", z0, "to", rows[-1, 3], "range", rows[:, 3].min(), rows[:, 3].max())
    print("posterior counts", out["z_hist_counts"].sum(), "edges", out["z_hist_edges"][[0, -1]], "abs", out["z_hist_edges_absolute"][[0, -1]], "relative_to", out["z_hist_relative_to"])


if __name__ == "__main__":
    main()
