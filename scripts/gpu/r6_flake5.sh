#!/bin/bash
# round 6: the two deterministic configurations of the f32 step with the split-form weight gradient: time and more repetitions
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_flake5; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "LIDAL_X_JOIN_BEFORE=bn" "LIDAL_PLAN_SIDE_F32=0" "A=1"; do
  env $cfg timeout 600 python bench.py --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-variants --no-roofline --no-families > $O/t.json 2> $O/t.err
  echo "$cfg: f32 ms_per_step $(python3 -c "import json; print(json.load(open('$O/t.json'))['ms_per_step'])")"
done
for cfg in "LIDAL_X_JOIN_BEFORE=bn" "LIDAL_PLAN_SIDE_F32=0"; do
  bad=0
  for r in 1 2 3 4 5 6; do
    env $cfg REPS=8 timeout 300 python3 scripts/exp/determinism_steps.py > $O/det.log 2>&1
    if grep -q "runs that differ" $O/det.log; then bad=$((bad+1)); fi
  done
  echo "$cfg: $bad of 6 rounds of 8 repetitions showed a difference"
done
