#!/bin/bash
# Sampler profile (run on the GPU box from the repo root): kernel trace + three PMC passes of
#   python scripts/bench_rjmcmc_device.py 65536 100
# PMC passes use --kernel-trace only (gpurun refuses --pmc with the sys/hip trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/rj
B=${1:-65536}; N=${2:-100}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/bench_rjmcmc_device.py $B $N > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/pmc_sq -o p -- python scripts/bench_rjmcmc_device.py $B $N > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python scripts/bench_rjmcmc_device.py $B $N > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python scripts/bench_rjmcmc_device.py $B $N > $OUT/write.log 2>&1
grep "B=" $OUT/kt.log | cut -c1-120
find $OUT -name "*counter_collection.csv" | head
