#!/bin/bash
# flake rate of scripts/exp/determinism_steps.py by configuration (REPS=16 repetitions of 6 steps, 3 rounds each)
for cfg in "A=1" "LIDAL_BN_FUSED=0" "LIDAL_PLAN_BRANCH_ROWS=0" "LIDAL_PLAN_SIDE_ROWS=0" "LIDAL_CONV_SPLIT=0" "LIDAL_BN_SUMS=0" "LIDAL_TAIL_SUMS_ROWS=0" "LIDAL_DEVOX_CELLS_AVG=0" "LIDAL_PLAN=0"; do
  bad=0
  for r in 1 2 3; do
    env $cfg REPS=16 python3 scripts/exp/determinism_steps.py 2>&1 | grep -q "runs that differ" && bad=$((bad+1))
  done
  echo "$cfg: $bad of 3 rounds showed a difference"
done
