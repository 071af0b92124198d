#!/bin/bash
# A/B builds of the library for same-box comparisons: scripts/build_ab.sh NAME [-Dflag ...]  ->  scripts/ab/NAME.so
# (measurement only: the product loads geobipy_amd/csrc/libgeobipy_amd.so, built by geobipy_amd/build.py with no -D flags)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scripts/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -pthread \
    "$@" geobipy_amd/csrc/gbp_fdem.hip -o scripts/ab/$name.so
echo built scripts/ab/$name.so
