#!/usr/bin/env python
"""The reference's loop-pair geometry moves (build container only) -> tests/golden/mcmc_geometry.npz.

The reference's Inference1D on skytem_glacial.csv row 30 with its skytem_options plus the keys that switch on moves of the loop
pair -- ``solve_transmitter_pitch``, ``solve_receiver_pitch``, ``solve_receiver_roll`` with their ``maximum_*_change`` and
``*_proposal_variance`` (system/Loop_pair.py:161-192, system/EmLoop.py:222-305; all False in the files the reference ships) -- run from
its seed on tests/golden/fake_gatdaem1d.py (forward numbers = this repository's oracle with the full geometry tuple, see
make_tdem_records.py: what is pinned is the reference's HOST logic -- which scalars move, in which order they consume the random
stream, their priors and posteriors -- never GA-AEM's numbers).  Per iteration: decision, layer count, misfit, the error levels and
the three angles; at the end the angle posteriors as the reference stores them.
Only inputs / outputs are stored (data), none of the reference's code.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fake_gatdaem1d                                  # noqa: E402
sys.modules["gatdaem1d"] = fake_gatdaem1d
from make_golden import REF, SUP, import_reference   # noqa: E402
from make_hdf_schema import Dataset, Group            # noqa: E402

import numpy as np   # noqa: E402

N_IT = 250
MOVES = dict(solve_transmitter_pitch=True, maximum_transmitter_pitch_change=4.0, transmitter_pitch_proposal_variance=0.3,
             solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.4,
             solve_receiver_roll=True, maximum_receiver_roll_change=3.0, receiver_roll_proposal_variance=0.25)


def main():
    import_reference()
    import h5py
    h5py.Group, h5py.File, h5py.Dataset = Group, Group, Dataset
    from geobipy import Inference1D, StatArray, TdemData, get_prng
    from geobipy.src.inversion import user_parameters as up
    opt = REF + "/documentation_source/source/supplementary/options_files/skytem_options"
    options = up.user_parameters.read(opt, data_directory=SUP)
    options["system_filename"] = [SUP + "/SkytemHM.stm", SUP + "/SkytemLM.stm"]
    options.update(n_markov_chains=N_IT, save_hdf5=True, interactive_plot=False, update_plot_every=5000, **MOVES)
    ds = TdemData.read_csv(SUP + "/skytem_glacial.csv", system=options["system_filename"])
    dp = ds.datapoint(30)
    for loop in (dp.loop_pair.transmitter, dp.loop_pair.receiver):
        if not hasattr(loop._orientation, "createHdf"):
            loop._orientation = StatArray(np.atleast_1d(np.asarray(loop._orientation)), "Orientation", dtype=np.int32)
    inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
    inf.initialize(dp)
    d = inf.datapoint
    st = inf.prng.bit_generator.state
    sv, inc, m64 = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
    rng_state = np.array([sv >> 64, sv & m64, inc >> 64, inc & m64, st["has_uint32"], st["uinteger"]], dtype=np.uint64)
    ang = lambda q: float(np.squeeze(q))
    tx, rx = d.transmitter, d.receiver
    out = dict(rng_state=rng_state, halfspace=float(inf.model.values[0]), data=np.asarray(d.data, dtype=np.float64).copy(), z=float(d.z[0]),
               prior0=float(inf.prior), like0=float(inf.likelihood), misfit0=float(inf.data_misfit),
               tx=np.array([ang(tx.x), ang(tx.y), ang(tx.z), ang(tx.pitch), ang(tx.roll), ang(tx.yaw)]),
               rx=np.array([ang(rx.x), ang(rx.y), ang(rx.z), ang(rx.pitch), ang(rx.roll), ang(rx.yaw)]),
               priors=np.array([[float(np.squeeze(q.prior.min)), float(np.squeeze(q.prior.max))] for q in (tx.pitch, rx.pitch, rx.roll)]),
               proposal_variances=np.array([float(np.squeeze(q.proposal.variance)) for q in (tx.pitch, rx.pitch, rx.roll)]),
               has_prior=np.array([[bool(getattr(l, n).hasPrior) for n in ("x", "y", "z", "pitch", "roll", "yaw")] for l in (tx, rx)]),
               offset_has_prior=np.array([bool(getattr(d.loop_pair, n).hasPrior) for n in ("x", "y", "z")]))
    rows = []
    for _ in range(N_IT):
        inf.accept_reject()
        inf.update()
        d = inf.datapoint
        rows.append(np.r_[float(bool(inf.accepted)), float(inf.model.nCells.item()), float(inf.data_misfit),
                          np.asarray(d.relative_error, dtype=np.float64), np.asarray(d.additive_error, dtype=np.float64),
                          ang(d.transmitter.pitch), ang(d.receiver.pitch), ang(d.receiver.roll), float(inf.prior), float(inf.likelihood)])
    rows = np.array(rows)
    d = inf.datapoint
    for name, q in (("tx_pitch", d.transmitter.pitch), ("rx_pitch", d.receiver.pitch), ("rx_roll", d.receiver.roll)):
        post = q.posterior
        out[name + "_hist_edges"] = np.asarray(post.mesh.edges, dtype=np.float64)
        out[name + "_hist_counts"] = np.asarray(post.values, dtype=np.int64)
        out[name + "_hist_relative_to"] = float(np.squeeze(post.mesh.relative_to))
    out["rows"] = rows
    out["move_options"] = np.array([[MOVES["maximum_transmitter_pitch_change"], MOVES["transmitter_pitch_proposal_variance"]],
                                    [MOVES["maximum_receiver_pitch_change"], MOVES["receiver_pitch_proposal_variance"]],
                                    [MOVES["maximum_receiver_roll_change"], MOVES["receiver_roll_proposal_variance"]]])
    np.savez_compressed(HERE + "/mcmc_geometry.npz", **out)
    print("accepted", int(rows[:, 0].sum()), "final k", rows[-1, 1], "misfit", rows[0, 2], "->", rows[-1, 2])
    print("angles", rows[-1, 7:10], "ranges", np.ptp(rows[:, 7:10], axis=0), "has_prior", out["has_prior"].tolist(), out["offset_has_prior"].tolist())
    print("bins", [out[n + "_hist_edges"].size for n in ("tx_pitch", "rx_pitch", "rx_roll")], [int(out[n + "_hist_counts"].sum()) for n in ("tx_pitch", "rx_pitch", "rx_roll")])


if __name__ == "__main__":
    main()
