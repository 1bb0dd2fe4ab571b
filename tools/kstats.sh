#!/bin/bash
# Register / scratch / code-size statistics of the k_env instantiations (cross-compiled, no GPU needed).  The headline
# kernel sits at 80 VGPRs (6 waves per SIMD) and ~58 KB of code (64 KB instruction cache): a change that spills or pushes
# the code past the cache costs far more than it saves -- check after every kernel edit.
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
cd $T && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include --save-temps -c $R/marl_llm_amd/csrc/swarm_env.hip -o $T/o.o 2>/dev/null
S=$T/swarm_env-hip-amdgcn-amd-amdhsa-gfx950.s
for K in "ILi64EfLb1ELb1ELb0E" "ILi64EfLb1ELb0ELb0E" "ILi64EfLb0ELb1ELb0E" "ILi32EfLb1ELb1ELb0E" "ILi32EfLb1ELb1ELb1E" "ILi8EfLb1ELb1ELb0E" "ILi128EfLb1ELb1ELb0E" "ILi256EfLb1ELb1ELb0E" "ILi256EfLb1ELb0ELb0E" "ILi64EdLb1ELb1ELb0E"; do
  L=$(grep -n "^_ZN12_GLOBAL__N_15k_env${K}EEvNS_2KPEPKviPT0_PfPhS5_:" $S | cut -d: -f1)
  E=$(awk -v L=$L 'NR>L && /\.end_amdhsa_kernel/{print NR; exit}' $S)
  echo "k_env<$K>" $(awk -v L=$L 'NR>L && /; (NumVgprs|ScratchSize|Occupancy|codeLenInByte)/{printf "%s ", $0; n++} n>=4{exit}' $S) "; spill instructions:" $(sed -n "${L},${E}p" $S | grep -c "Folded Spill\|Folded Reload")
done
[ -n "$KEEP_ASM" ] && cp $S $KEEP_ASM
rm -rf $T
