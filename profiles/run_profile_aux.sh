#!/bin/bash
# kernel-trace + stats for the Jacobian and TDEM scripts (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/aux
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/jac -o kt -- python scripts/bench_jacobian.py > $OUT/jac.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tdem -o kt -- python scripts/bench_tdem.py > $OUT/tdem.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $OUT/jac_pmc -o p -- python scripts/bench_jacobian.py > $OUT/jac_pmc.log 2>&1
tail -n 3 $OUT/jac.log; tail -n 3 $OUT/tdem.log
