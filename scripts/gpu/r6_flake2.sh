#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_flake2; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "A=1" "LIDAL_X_WGRAD_ARENA=1" "LIDAL_X_JOIN_AFTER_WGRAD=1" "LIDAL_X_SPLIT_DENSE=0" "LIDAL_X_SPLIT_CONV=0" "LIDAL_PLAN_SIDE_MIN_ROWS=150000" "LIDAL_PLAN_SIDE_ROWS=150000" "LIDAL_F32_SPLIT=0"; do
  bad=0
  for r in 1 2; do
    env $cfg REPS=6 timeout 300 python3 scripts/exp/determinism_steps.py > $O/det.log 2>&1
    if grep -q "runs that differ" $O/det.log; then bad=$((bad+1)); fi
    grep -m1 "gradients of [1-9]" $O/det.log | cut -c1-120
  done
  echo "$cfg: $bad of 2 rounds showed a difference"
done
