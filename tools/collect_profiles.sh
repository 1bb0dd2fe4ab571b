#!/bin/bash
# One GPU call that regenerates the judged artefacts under gpurun_out/final/ (copy them into profiles/<round>/):
#   bench_n1.json                 the default `python3 bench.py` line (cpu_baseline on all host cores, other_configs)
#   kernel_stats.csv              rocprofv3 --kernel-trace --stats of the same command (per-kernel average duration)
#   kernel_stats_<shape>.csv      the same for every other shape BASELINE.json names (32x1024, 256x4096, 64x32768,
#                                 scatter state, reference fig shapes through other_configs is covered by bench_n1.json)
#   final_pmc_summary.json        HBM traffic and SQ counters per launch (separate --pmc passes), stamped with the sha256
#                                 of csrc/swarm_env.hip so bench.py only quotes it for the binary it was taken on
#   cumulative_time.txt / cumulative_valu.txt   per-segment profile of the FINAL binary (early-exit runs)
#   rehearse_gpus2_one_gpu.json   `python3 bench.py --gpus 2 --rehearse-one-gpu` (spawns its own ranks)
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh
# Every profiled program is `python3 <script>` directly after `--` (no env/bash hop: the profiler has initialised the GPU).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.json; echo
kt() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs "$@" > $OUT/kt_$name.json 2> $OUT/kt_$name.log
  cp $(find $OUT/kt_$name -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$name.csv && head -2 $OUT/kernel_stats_$name.csv | cut -c1-200
  rm -rf $OUT/kt_$name
}
kt headline
kt n32_e1024 --agents 32 --envs 1024
kt n256_e4096 --agents 256 --envs 4096 --steps 50
kt n64_e32768 --envs 32768 --steps 50
kt scatter --state scatter
echo "kernel stats done"
CSVS=""
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS_ATOMIC" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  T=$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $G -d $OUT/pmc_$T -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --prewarm-ms 0 > /dev/null 2>$OUT/pmc_$T.log
  F=$(find $OUT/pmc_$T -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && CSVS="$CSVS $F"
  echo "pmc pass $T done"
done
python3 $R/tools/pmc_summary.py $OUT/final_pmc_summary.json 20 "assembly env, 64 agents x 4096 envs per GPU, assembled state" $CSVS > $OUT/pmc_print.txt
rm -rf $OUT/pmc_*/
python3 $R/tools/phase_profile.py 64 4096 > $OUT/per_role_stamps.txt 2>&1      # needs marl_llm_amd/lib/libswarmenv_stamps.so (build_lib(stamps=True))
python3 $R/tools/phase_profile.py 256 4096 > $OUT/per_role_stamps_n256_e4096.txt 2>&1
python3 $R/tools/ablate.py --cumulative > $OUT/cumulative_time.txt 2>&1
echo "cumulative time done"
export ABLATE_STEPS=10
rocprofv3 --pmc SQ_INSTS_VALU -d $OUT/cumpmc -o p --output-format csv -- python3 $R/tools/ablate.py --cumulative > /dev/null 2>$OUT/cumpmc.log
python3 $R/tools/ablate_pmc_cum.py $(find $OUT/cumpmc -name "*counter_collection.csv" | head -1) 10 > $OUT/cumulative_valu.txt
rm -rf $OUT/cumpmc
echo "cumulative valu done"
python3 $R/tools/rollout_bench.py > $OUT/rollout_bench_n64_e4096.txt 2>&1
python3 $R/tools/stress_consistency.py > $OUT/stress_consistency.txt 2>&1
tail -1 $OUT/stress_consistency.txt
cd $R && python3 bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 --envs 512 --no-other-configs > $OUT/rehearse_gpus2_one_gpu.json 2> $OUT/rehearse.err
cat $OUT/rehearse_gpus2_one_gpu.json | cut -c1-300
ls $OUT
