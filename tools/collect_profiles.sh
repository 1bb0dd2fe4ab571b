#!/bin/bash
# One GPU call that regenerates the judged artefacts under gpurun_out/final/ (copy them into profiles/<round>/):
#   bench_n1.json            the default `python bench.py` line (with cpu_baseline)
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of the same command (per-kernel average duration)
#   pmc_summary.json         HBM traffic and SQ counters per launch (separate --pmc passes)
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_n1.json
cat $OUT/bench_n1.json
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > /dev/null 2>$OUT/kt.log
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -3 $OUT/kernel_stats.csv
CSVS=""
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY"; do
  T=$(echo $G | tr ' ' '_')
  rocprofv3 --pmc $G -d $OUT/pmc_$T -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>$OUT/pmc_$T.log
  CSVS="$CSVS $(find $OUT/pmc_$T -name '*counter_collection.csv' | head -1)"
done
python3 $R/tools/pmc_summary.py $OUT/pmc_summary.json 20 "assembly env, 64 agents x 4096 envs per GPU, assembled state" $CSVS
rm -rf $OUT/kt $OUT/pmc_*/ 
