#!/bin/bash
# A/B timing of step-kernel builds on ONE box (run-to-run and box-to-box spread is ~2 %, so variants are only comparable
# within a call): bash tools/ab.sh libA.so libB.so ...   -> kernel us of each, three alternating rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for L in "$@"; do
    export SWARM_LIB=$R/$L
    python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 300 > /tmp/ab.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$L', round(d['roofline']['kernel_us'],2))"
  done
done
