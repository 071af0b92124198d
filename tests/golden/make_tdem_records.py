#!/usr/bin/env python
"""The reference's OWN time-domain host logic, run in the build container -> tests/golden/hdf_schema_tdem.json, mcmc_trace_tdem.npz.

gatdaem1d is absent, so the reference's TdemDataPoint / Tempest_datapoint / Inference1D run here on the stand-in
tests/golden/fake_gatdaem1d.py, whose physics is THIS repository's oracle (oracle/tdem_oracle.py).  What is recorded therefore
pins the reference's host logic around the forward operator, never GA-AEM's numbers:

  hdf_schema_tdem.json   the tree Inference1D.createHdf / writeHdf (inversion/Inference1D.py:1002-1090) build for a SkyTEM (two
                         systems, Z) and a Tempest (X and Z, primary field, receiver pitch) data point -- through
                         TdemDataPoint.createHdf / writeHdf (data/datapoint/TdemDataPoint.py:603-645) and
                         Tempest_datapoint.createHdf (Tempest_datapoint.py:566-586) -- recorded with the in-memory h5py stand-in of
                         make_hdf_schema.py: groups, datasets, shapes, dtypes, fill values, attributes, small values;
  mcmc_trace_tempest.npz the same for tempest_glacial.csv row 30 with tempest_options (200 iterations): decision, layer count, misfit, the
                         two relative levels, the two additive-error MULTIPLIERS (Tempest_datapoint.py:94-104, 339-341), prior and
                         likelihood, with the total-field model's inputs (secondary / primary / predicted primary field, per-channel
                         additive errors) -- what a host twin of Tempest_datapoint has to reproduce;
  mcmc_trace_tdem.npz    a seeded run of the reference's sampler on skytem_glacial.csv row 30 with skytem_options: per iteration the
                         decision, layer count, misfit and the two sets of error levels (proposed JOINTLY, DataPoint.perturb
                         :531-573), the starting half-space, and the inputs (data, geometry) -- what geobipy_amd's host
                         Inference1D on a TdemDataPoint with the same oracle as its engine has to reproduce decision by decision.
Only names / shapes / numbers are stored (data), none of the reference's code.
"""
import json
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

N_TRACE = 300


def run(kind, n_it, trace):
    from geobipy import Inference1D, StatArray, get_prng
    from geobipy.src.inversion import user_parameters as up
    opt = REF + "/documentation_source/source/supplementary/options_files/" + ("skytem_options" if kind == "skytem" else "tempest_options")
    options = up.user_parameters.read(opt, data_directory=SUP)
    if kind == "skytem":
        from geobipy import TdemData as Data
        options["system_filename"] = [SUP + "/SkytemHM.stm", SUP + "/SkytemLM.stm"]
        csv = SUP + "/skytem_glacial.csv"
    else:
        from geobipy import TempestData as Data
        options["system_filename"] = SUP + "/tempest.stm"
        csv = SUP + "/tempest_glacial.csv"
    options.update(n_markov_chains=n_it, save_hdf5=True, interactive_plot=False, update_plot_every=5000)
    if os.environ.get("GBP_SCHEMA_VARIANT") == "pitch":      # the same Tempest run with the receiver pitch sampled -> hdf_schema_tempest_pitch.json
        options.update(solve_receiver_pitch=True, maximum_receiver_pitch_change=5.0, receiver_pitch_proposal_variance=0.3)
    ds = Data.read_csv(csv, system=options["system_filename"]) if kind == "skytem" else Data.read_csv(csv, options["system_filename"])
    if kind == "tempest":        # (the dataset's error levels start at 0, which Tempest_datapoint refuses: the example scripts set them first)
        ds.relative_error = np.tile(np.atleast_1d(options["initial_relative_error"]).astype(float), (ds.nPoints, 1))
        ds.additive_error = np.tile(np.atleast_1d(options["initial_additive_error"]).astype(float), (ds.nPoints, 1))
    dp = ds.datapoint(30)
    # (EmLoop.__getitem__, system/EmLoop.py:63, leaves a numpy scalar in _orientation under this container's numpy, which
    # EmLoop.createHdf :429 cannot write: put the one-element StatArray back -- object state only, nothing of the reference is edited)
    for loop in (dp.loop_pair.transmitter, dp.loop_pair.receiver):
        if not hasattr(loop._orientation, "createHdf"):
            loop._orientation = StatArray(np.atleast_1d(np.asarray(loop._orientation)), "Orientation", dtype=np.int32)
    inf = Inference1D(prng=get_prng(seed=options["seed"]), world=None, **options)
    inf.initialize(dp)
    root = Group("/")
    fid = np.sort(np.asarray(ds.fiducial)[[29, 30, 31]])
    inf.createHdf(root, add_axis=fid)
    StatArray(fid).writeHdf(root, "data/fiducial")
    d = inf.datapoint
    rec = dict(halfspace=float(inf.model.values[0]), data=np.asarray(d.data, dtype=np.float64).copy(), z=float(d.z[0]),
               misfit0=float(inf.data_misfit), prior0=float(inf.prior), like0=float(inf.likelihood))
    if kind == "tempest":            # the total-field model's inputs (Tempest_datapoint.py:106-176) and the generator state
        st = inf.prng.bit_generator.state
        sv, inc, m64 = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
        rec.update(rng_state=np.array([sv >> 64, sv & m64, inc >> 64, inc & m64, st["has_uint32"], st["uinteger"]], dtype=np.uint64),
                   secondary_field=np.asarray(d.secondary_field, dtype=np.float64).copy(), primary_field=np.asarray(d.primary_field, dtype=np.float64).copy(),
                   predicted_primary_field=np.asarray(d.predicted_primary_field, dtype=np.float64).copy(),
                   additive_error=np.asarray(d.additive_error, dtype=np.float64).copy(),
                   multiplier0=np.asarray(d.additive_error_multiplier, dtype=np.float64).copy(),
                   relative_error0=np.asarray(d.relative_error, dtype=np.float64).copy(),
                   offset=np.array([float(np.squeeze(d.loop_pair.x)), float(np.squeeze(d.loop_pair.y)), float(np.squeeze(d.loop_pair.z))]),
                   tx_z=float(np.squeeze(d.transmitter.z)))
    rows = []
    for _ in range(n_it):
        inf.accept_reject()
        inf.update()
        if trace:
            d = inf.datapoint
            lev = np.asarray(d.additive_error_multiplier, dtype=np.float64) if kind == "tempest" else np.asarray(d.additive_error, dtype=np.float64)
            rows.append(np.r_[float(bool(inf.accepted)), float(inf.model.nCells.item()), float(inf.data_misfit),
                              np.asarray(d.relative_error, dtype=np.float64), lev, float(inf.prior), float(inf.likelihood),
                              float(np.squeeze(d.receiver.pitch))])
    inf.writeHdf(root, index=1)
    tree = {}
    root.walk(tree)
    keep = (int, float, bool, str, type(None))
    meta = {"sounding": os.path.basename(csv) + " row 30", "iterations": n_it, "index": 1, "n_points": 3, "fiducials": [float(x) for x in fid],
            "seed": str(options["seed"]), "iteration": int(inf.iteration), "k": int(inf.model.nCells.item()),
            "options": {k: (v if isinstance(v, keep) else [float(x) for x in np.atleast_1d(v)]) for k, v in options.items()
                        if k not in ("seed", "system_filename", "data_filename", "data_directory")
                        and (isinstance(v, keep) or (isinstance(v, (list, np.ndarray)) and all(isinstance(x, (int, float, np.floating)) for x in np.atleast_1d(v))))},
            "note": "tree recorded from the reference's own createHdf / writeHdf on tests/golden/fake_gatdaem1d.py (forward values are this "
                    "repository's oracle, not GA-AEM's) through an in-memory stand-in for h5py"}
    return meta, tree, rec, np.array(rows)


def main():
    import_reference()
    import h5py
    h5py.Group, h5py.File, h5py.Dataset = Group, Group, Dataset
    out = {}
    if os.environ.get("GBP_SCHEMA_VARIANT") == "pitch":
        meta, tree, rec, rows = run("tempest", 60, True)
        json.dump({"meta": dict(meta, variant="pitch"), "tree": tree}, open(HERE + "/hdf_schema_tempest_pitch.json", "w"), indent=0, sort_keys=True)
        np.savez_compressed(HERE + "/mcmc_trace_tempest_pitch.npz", rows=rows, **{k: np.asarray(v) for k, v in rec.items()})
        print("tempest + receiver pitch:", len(tree), "entries; accepted", int(rows[:, 0].sum()), "pitch", rows[-1, -1])
        return
    # (the container trees come from the 300- / 60-iteration runs; the Tempest trace from a run of its own, 200 iterations)
    for kind, n_it, trace, schema in (("skytem", N_TRACE, True, True), ("tempest", 60, False, True), ("tempest", 200, True, False)):
        meta, tree, rec, rows = run(kind, n_it, trace)
        if schema:
            out[kind] = {"meta": meta, "tree": tree}
        print(kind, len(tree), "entries;", sum(1 for v in tree.values() if v["kind"] == "dataset"), "datasets; k =", meta["k"])
        if trace:
            np.savez_compressed(HERE + ("/mcmc_trace_tdem.npz" if kind == "skytem" else "/mcmc_trace_tempest.npz"), rows=rows,
                                **{k: np.asarray(v) for k, v in rec.items()})
            print("trace", rows.shape, "accepted", int(rows[:, 0].sum()), "final k", rows[-1, 1], "misfit", rows[-1, 2])
    json.dump(out, open(HERE + "/hdf_schema_tdem.json", "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
