#!/bin/bash
# flake rate after the change (5 rounds of 16 repetitions), planned and per-operator
for cfg in "A=1" "LIDAL_PLAN=0" "LIDAL_TAIL_SUMS_ROWS=100000"; do
  bad=0
  for r in 1 2 3 4 5; do
    env $cfg REPS=16 python3 scripts/exp/determinism_steps.py 2>&1 | grep -q "runs that differ" && bad=$((bad+1))
  done
  echo "$cfg: $bad of 5 rounds showed a difference"
done
