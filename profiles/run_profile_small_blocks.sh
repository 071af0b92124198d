#!/bin/bash
# Sampler in the small-block regime (run on the GPU box from the repo root): kernel trace + stats of
#   the lock-step driver at 8 192 chains (fused physics launches)  -> gpurun_out/prof/small/lock8192
#   the persistent kernel at 1 024 chains x 2 000 iterations         -> gpurun_out/prof/small/pers1024
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof/small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lock8192 -o kt -- python scripts/bench_rjmcmc_device.py 8192 300 > $OUT/lock8192.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pers1024 -o kt -- python scripts/bench_rjmcmc_device.py 1024 2000 > $OUT/pers1024.log 2>&1
grep "B=" $OUT/lock8192.log $OUT/pers1024.log | cut -c1-160
python scripts/trace_timeline.py $(find $OUT/lock8192 -name "*kernel_trace.csv") 300 > $OUT/timeline_lock8192.txt; cat $OUT/timeline_lock8192.txt | head -12
