#!/bin/bash
# Round-6 profiles (the passes of rounds 3 - 5, plus an un-profiled run of the case first: plain.log) (run on the GPU box from the repo root):  bash profiles/run_profile_r6.sh case [case ...]
# per case of scripts/prof_case.py: kernel trace + stats, then three PMC passes (SQ issue counters / FETCH_SIZE / WRITE_SIZE), each in
# its own run with --kernel-trace only (gpurun refuses --pmc together with the sys / hip trace domains) -> gpurun_out/prof/<case>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
for CASE in "$@"; do
  OUT=$R/gpurun_out/prof/$CASE
  mkdir -p $OUT
  python scripts/prof_case.py $CASE > $OUT/plain.log 2>&1      # the un-profiled rate of the same case on the same box
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python scripts/prof_case.py $CASE > $OUT/kt.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/pmc_sq -o p -- python scripts/prof_case.py $CASE > $OUT/sq.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python scripts/prof_case.py $CASE > $OUT/fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python scripts/prof_case.py $CASE > $OUT/write.log 2>&1
  grep CASE $OUT/kt.log
done
