#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_flake3; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "LIDAL_X_SPLIT_APPLY=0" "LIDAL_BN_FUSED=0" "LIDAL_CONV_SPLIT=0" "DTYPE=bf16"; do
  bad=0
  for r in 1 2 3; do
    env $cfg REPS=6 timeout 300 python3 scripts/exp/determinism_steps.py > $O/det.log 2>&1
    if grep -q "runs that differ" $O/det.log; then bad=$((bad+1)); fi
    grep -m1 "gradients of [1-9]" $O/det.log | cut -c1-120
  done
  echo "$cfg: $bad of 3 rounds showed a difference"
done
