#!/bin/bash
# static instruction mix of one k_env instantiation (default <64, float, true>) -- diagnostics only
set -e
OUT=${2:-/tmp/swarm_env.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o "$OUT" marl_llm_amd/csrc/swarm_env.hip
SYM=${1:-_ZN12_GLOBAL__N_15k_envILi64EfLb1EEEvNS_2KPEPKviPT0_PfPhS5_}
awk -v sym="$SYM" '$0 ~ "^"sym":" {on=1} on && /s_endpgm/ {print; on=0} on {print}' "$OUT" > /tmp/kfn.s
echo "lines: $(wc -l < /tmp/kfn.s)"
echo "VALU: $(grep -cE '^\s+v_' /tmp/kfn.s)  SALU: $(grep -cE '^\s+s_' /tmp/kfn.s)  LDS: $(grep -cE '^\s+ds_' /tmp/kfn.s)  global: $(grep -cE '^\s+global_' /tmp/kfn.s) scratch: $(grep -cE '^\s+scratch_' /tmp/kfn.s)"
grep -E "^\s+\.(vgpr_count|sgpr_count|lds_size)|NumVgprs|NumSgprs|ScratchSize|Occupancy" "$OUT" | head -0
awk -v sym="$SYM" '$0 ~ "\\.name:.*"sym {on=1} on && /(vgpr_count|sgpr_count|private_segment_fixed_size)/ {print} on && /\.wavefront_size/ {on=0}' "$OUT"
