#!/bin/bash
# Everything profiles/r6 holds, in one call on the GPU box (from the repo root):  bash profiles/collect_r6.sh
# -> gpurun_out/profiles_r6/ (summaries, kernel stats, chain trace); the bulky rocprofv3 CSVs stay on the box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p profiles/r6 gpurun_out/profiles_r6
bash profiles/run_profile.sh r6bench > gpurun_out/profiles_r6/run_r6bench.log 2>&1
python profiles/summarise_bench.py r6bench r6 > /dev/null 2> gpurun_out/profiles_r6/summarise_bench.err
cp $(find gpurun_out/prof/r6bench/kt -name "*kernel_stats.csv" | head -1) profiles/r6/kernel_stats_bench_65536x10x8.csv
bash profiles/run_profile_r6.sh rjmcmc_8192 rjmcmc_1024 jacobian_headline tdem_config4 config2 > gpurun_out/profiles_r6/run_cases.log 2>&1
for c in rjmcmc_8192 rjmcmc_1024 jacobian_headline tdem_config4 config2; do
  python profiles/summarise_case.py $c r6 > /dev/null 2>> gpurun_out/profiles_r6/summarise_case.err
done
python scripts/trace_chain.py $(find gpurun_out/prof/rjmcmc_8192/kt -name "*kernel_trace.csv" | head -1) > profiles/r6/chain_8192.txt 2>> gpurun_out/profiles_r6/summarise_case.err
cp $(find gpurun_out/prof/rjmcmc_8192/kt -name "*kernel_stats.csv" | head -1) profiles/r6/kernel_stats_rjmcmc_8192.csv
cp -r profiles/r6/. gpurun_out/profiles_r6/
rm -rf gpurun_out/prof
ls -la gpurun_out/profiles_r6
