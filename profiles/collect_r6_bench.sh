#!/bin/bash
# The headline profile of profiles/r6 alone (kernel trace + stats and the PMC passes of profiles/run_profile.sh) -> gpurun_out/profiles_r6/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p profiles/r6 gpurun_out/profiles_r6
bash profiles/run_profile.sh r6bench > gpurun_out/profiles_r6/run_r6bench.log 2>&1
python profiles/summarise_bench.py r6bench r6 > /dev/null 2> gpurun_out/profiles_r6/summarise_bench.err
cp $(find gpurun_out/prof/r6bench/kt -name "*kernel_stats.csv" | head -1) profiles/r6/kernel_stats_bench_65536x10x8.csv
cp profiles/r6/summary_bench_65536x10x8.json profiles/r6/kernel_stats_bench_65536x10x8.csv gpurun_out/profiles_r6/
rm -rf gpurun_out/prof
