#!/usr/bin/env python
"""gatdaem1d numbers the reference tree holds in its RENDERED gallery -> tests/golden/tdem_doc_pins.npz  (data only).

gatdaem1d (GA-AEM) is absent from /root/reference and cannot be built here, but the sphinx-gallery pages under
/root/reference/docs/_sources/examples/Datapoints were rendered by the reference's author WITH it, and print its outputs for
fully specified inputs.  This script (run in the build container; the reference never travels) parses those printed numbers and
stores them next to the inputs that produced them:

  Tempest   plot_tempest_datapoint.rst.txt  (source: documentation_source/source/examples/Datapoints/plot_tempest_datapoint.py)
      row 0 of supplementary/data/tempest_saline_clay.csv, tempest.stm, model sigma = logspace(-3, 3, 30) S/m on edges
      linspace(0, 350, 31) (last = inf)                                                        (.py:62-64)
      J  = tdp.sensitivity(mod)        30 channels (15 X, 15 Z) x 30 layers, d pred / d ln sigma  (.rst:192-431; gaTdem1dsen,
                                       TD/tdem1d.py:125-154 = sigma * gatdaem1d.derivative(CONDUCTIVITYDERIVATIVE, layer))
      J2 = tdp.fm_dlogc(mod); tdp.sensitivity_matrix   the same quantity from gatdaem1d.fm_dlogc  (.rst:441-680; ga_fm_dlogc :98-123)
      relative error [0.001, 0.001], additive error 30 values (.py:117-121), then
      logL = tdp.likelihood(log=True) = -36389.6500813217, chi2 = tdp.data_misfit() = 72940.71365767403   (.rst:736-737)
      best half-space 0.01830738 S/m                                                            (.rst:827)
      primary field printed [34.27253219 17.55503397] = the PX, PZ columns of the CSV row: the page was rendered by the tree's version.
  SkyTEM    plot_skytem_datapoint.rst.txt   (source plot_skytem_datapoint.py)
      row 0 of skytem_saline_clay.csv, SkytemHM.stm + SkytemLM.stm, model sigma = [500, 20] S/m, edges [0, 75, inf]   (.py:62-63)
      relative error [0.05, 0.05], additive error [1e-14, 1e-13]                                  (.py:96-97)
      logL = -320327.7331520335, chi2 = 643134.8665683016                                         (.rst:281-282)
      best half-space 0.01047616 S/m (.rst:345) and its chi2 19656.315144677585                   (.rst:368)

The printed arrays carry 9 significant digits (numpy's default print precision), the scalars 16-17.
"""
import os
import re
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DOC = REF + "/docs/_sources/examples/Datapoints/"
SUP = REF + "/documentation_source/source/supplementary/data/"

NUM = r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?"


def printed_matrix(text, label):
    """The numpy-printed 2-D array following ``label [[`` in a script-out block."""
    m = re.search(r"^\s*" + re.escape(label) + r" \[\[", text, re.M)
    assert m, label
    body = text[m.end() - 2:]
    end = body.index("]]") + 2
    rows = re.findall(r"\[([^\[\]]*)\]", body[:end])
    return np.array([[float(x) for x in r.split()] for r in rows])


def main():
    out = {}
    # ---------------------------------------------------------------- Tempest
    t = open(DOC + "plot_tempest_datapoint.rst.txt").read()
    J, J2 = printed_matrix(t, "J"), printed_matrix(t, "new J")
    assert J.shape == (30, 30) and J2.shape == (30, 30)
    prim = np.array([float(x) for x in re.search(r"^\s*primary \[(.*?)\]", t, re.M).group(1).split()])
    m = re.search(r"code-block:: none\s*\n\s*\n\s*(-\d+\.\d+)\s*\n\s*(\d+\.\d+)\s*\n", t)
    logl, chi2 = float(m.group(1)), float(m.group(2))
    best = float(re.search(r"Best half space conductivity is \[(" + NUM + r")\]", t).group(1))
    src = open(REF + "/documentation_source/source/examples/Datapoints/plot_tempest_datapoint.py").read()
    add = np.array([float(x) for x in re.findall(r"0\.\d{6}", src[src.index("tdp.additive_error = np.hstack"):src.index("tdp.predictedData.prior")])])
    assert add.size == 30 and "tdp.relative_error = np.r_[0.001, 0.001]" in src and "np.logspace(-3, 3, 30)" in src and "np.linspace(0, 350, 31)" in src
    hdr = open(SUP + "tempest_saline_clay.csv").readline().strip().split(",")
    row = np.loadtxt(SUP + "tempest_saline_clay.csv", delimiter=",", skiprows=1, max_rows=1)
    col = {k: row[i] for i, k in enumerate(hdr)}
    assert abs(col["PX"] - prim[0]) < 5e-9 and abs(col["PZ"] - prim[1]) < 5e-9         # rendered by the tree's version
    out.update(tempest_J=J, tempest_J_fm_dlogc=J2, tempest_logl=logl, tempest_chi2=chi2, tempest_best_halfspace=best,
               tempest_sigma=np.logspace(-3, 3, 30), tempest_edges=np.linspace(0, 350, 31)[1:-1],
               tempest_relative_error=np.r_[0.001, 0.001], tempest_additive_error=add,
               tempest_secondary=row[17:47], tempest_primary=row[15:17], tempest_height=col["Height"],
               tempest_geometry=np.array([col["Height"], col["tx_roll"], col["tx_pitch"], col["tx_yaw"], col["txrx_dx"], col["txrx_dy"],
                                          col["txrx_dz"], col["rx_roll"], col["rx_pitch"], col["rx_yaw"]]))
    # ---------------------------------------------------------------- SkyTEM
    t = open(DOC + "plot_skytem_datapoint.rst.txt").read()
    m = re.search(r"code-block:: none\s*\n\s*\n\s*(-\d+\.\d+)\s*\n\s*(\d+\.\d+)\s*\n", t)
    logl, chi2 = float(m.group(1)), float(m.group(2))
    best = float(re.search(r"Best half space conductivity is \[(" + NUM + r")\]", t).group(1))
    chi2_best = float(re.search(r"code-block:: none\s*\n\s*\n\s*(\d+\.\d+)\s*\n\s*\n", t[t.index("Best half space"):]).group(1))
    src = open(REF + "/documentation_source/source/examples/Datapoints/plot_skytem_datapoint.py").read()
    assert "np.r_[500.0, 20.0]" in src and "edges=np.r_[0, 75.0, np.inf]" in src
    assert "tdp.relative_error = np.r_[0.05, 0.05]" in src and "tdp.additive_error = np.r_[1e-14, 1e-13]" in src
    hdr = open(SUP + "skytem_saline_clay.csv").readline().strip().split(",")
    row = np.loadtxt(SUP + "skytem_saline_clay.csv", delimiter=",", skiprows=1, max_rows=1)
    col = {k: row[i] for i, k in enumerate(hdr)}
    out.update(skytem_logl=logl, skytem_chi2=chi2, skytem_best_halfspace=best, skytem_chi2_best_halfspace=chi2_best,
               skytem_sigma=np.r_[500.0, 20.0], skytem_edges=np.r_[75.0], skytem_relative_error=np.r_[0.05, 0.05],
               skytem_additive_error=np.r_[1e-14, 1e-13], skytem_data=row[15:60], skytem_height=col["Height"],
               skytem_geometry=np.array([col["Height"], col["tx_roll"], col["tx_pitch"], col["tx_yaw"], col["txrx_dx"], col["txrx_dy"],
                                         col["txrx_dz"], col["rx_roll"], col["rx_pitch"], col["rx_yaw"]]))
    np.savez_compressed(os.path.join(HERE, "tdem_doc_pins.npz"), **out)
    for k, v in out.items():
        print(k, np.shape(v), v if np.size(v) <= 2 else "")


if __name__ == "__main__":
    main()
