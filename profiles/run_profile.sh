#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun from the repo root):
#   bash profiles/run_profile.sh <tag> [bench args]
# 1. kernel trace + stats (CSV)            -> gpurun_out/prof/<tag>/kt
# 2. PMC pass A: FETCH_SIZE (TCC, 3 slots) -> gpurun_out/prof/<tag>/pmc_fetch
# 3. PMC pass B: WRITE_SIZE                -> gpurun_out/prof/<tag>/pmc_write
# 4. PMC pass C: SQ issue/wait counters    -> gpurun_out/prof/<tag>/pmc_sq
# PMC passes never combine with --sys-trace etc. (gpurun refuses that); they use --kernel-trace only.
TAG=${1:-r1}; shift
ARGS=${@:-"--steps 20 --warmup 3 --no-cpu-baseline --no-windowed --no-rjmcmc --no-extras"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py $ARGS > $OUT/bench_kt.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o p -- python bench.py $ARGS > $OUT/bench_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VALU -d $OUT/pmc_sq2 -o p -- python bench.py $ARGS > $OUT/bench_sq2.log 2>&1
find $OUT -name "*.csv" | head -40
rocprofv3 -L > $OUT/counters_list.txt 2>&1
