#!/usr/bin/env python
"""A results container written by the DEVICE path, as a fixture (run on a GPU box: gpurun -- python tests/golden/make_device_h5.py):
survey.infer on eight soundings of the Resolve wedge file (tests/golden/resolve_options_small, 400 iterations, seed 11) with
container="hdf5" -> gpurun_out/device_survey/0.0.h5 (the file geobipy_amd.h5lite wrote from the device sampler's rows: LineSpec tree,
hit maps as deflated chunks, traces) and summary.npz (what the same call returned).  Copied to tests/golden/device_survey_0.0.h5 /
device_survey_summary.npz; tests/test_hdf5_file.py has the reference's own readers open the file in the build container."""
import os, shutil, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from geobipy_amd import survey
o = survey.read_options(os.path.join(HERE, "resolve_options_small"))
ds = survey.FdemData.read_csv(o["data_filename"], o["system_filename"]).subset(np.arange(0, 79, 10))
d = tempfile.mkdtemp()
res = survey.infer(os.path.join(HERE, "resolve_options_small"), data=ds, n_markov_chains=400, burn_in_min_iterations=150, check_every=100, seed=11,
                   results_directory=d, container="hdf5", output=os.path.join(d, "summary.npz"))
out = os.path.join(ROOT, "gpurun_out", "device_survey")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(d, "0.0.h5"), out)
shutil.copy(os.path.join(d, "summary.npz"), out)
print(sorted(os.listdir(d)), os.path.getsize(os.path.join(d, "0.0.h5")), "status", res["status"].tolist(), "k", res["best_n_layers"].tolist())
