#!/bin/bash
# round 6: the f32 step with the split weight gradient beside it is not bit-reproducible -- which configuration, which victim?
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_flake; mkdir -p $O
cd $GRAFT_REPO_ROOT
REPS=6 timeout 600 python scripts/exp/victim/run_f64.py 2>&1 | grep -v amdgpu.ids | tee $O/victim4.txt
for cfg in "A=1" "LIDAL_PLAN_SIDE_F32=0" "LIDAL_F32_SPLIT_TRAIN=0" "LIDAL_PLAN_BRANCH_ROWS=0" "LIDAL_PLAN=0" "NO_DROPOUT=1"; do
  bad=0
  for r in 1 2 3; do
    env $cfg REPS=8 timeout 300 python3 scripts/exp/determinism_steps.py > $O/det.log 2>&1
    if grep -q "runs that differ" $O/det.log; then bad=$((bad+1)); grep -m3 "step [0-9]: loss differs\|gradients that agree\|max |d|" $O/det.log | cut -c1-400 > $O/det_$(echo $cfg | tr '=' '_')_$r.txt; fi
  done
  echo "$cfg: $bad of 3 rounds showed a difference"
done
cat $O/det_A_1_1.txt 2>/dev/null | cut -c1-600
