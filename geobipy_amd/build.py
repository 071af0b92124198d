"""Build the gfx950 HIP library in-tree (geobipy_amd/csrc/libgeobipy_amd.so)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["gbp_fdem.hip"]
HEADERS = ["gbp_math.h", "gbp_math_tables.h", "gbp_fdem_point.h", "gbp_fdem_tables.h", "gbp_rjmcmc.h", "gbp_tdem.h", "gbp_hostpack.h", "gbp_hitmap.h", "../../include/geobipy_amd.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 library")
    return exe


def build_native(force=False, verbose=False):
    out = os.path.join(CSRC, "libgeobipy_amd.so")
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out
