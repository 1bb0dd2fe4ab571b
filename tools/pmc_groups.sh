#!/bin/bash
# Collect SQ / memory counters of the headline bench in separate rocprofv3 --pmc passes (one counter group per pass) and
# print the per-launch mean of every counter over the timed k_env<...,true> launches.
# Usage (GPU box, repo root): bash tools/pmc_groups.sh OUTNAME "GROUP 1 COUNTERS" "GROUP 2 COUNTERS" ...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CSVS=""
for G in "$@"; do
  T=$(echo $G | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $G -d $OUT/pmc_$T -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > /dev/null 2>$OUT/pmc_$T.log || { tail -5 $OUT/pmc_$T.log; continue; }
  F=$(find $OUT/pmc_$T -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && CSVS="$CSVS $F"
done
python3 $R/tools/pmc_summary.py $OUT/summary.json 20 "assembly env, 64 agents x 4096 envs per GPU, assembled state" $CSVS
rm -rf $OUT/pmc_*/
