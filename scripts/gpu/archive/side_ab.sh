#!/bin/bash
# the plan's side streams again with the process bound: weight gradients (1), shortcut branches (2)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "X=1" "LIDAL_PLAN_SIDE_ROWS=0" "LIDAL_PLAN_BRANCH_ROWS=0" "LIDAL_PLAN_SIDE_ROWS=0 LIDAL_PLAN_BRANCH_ROWS=0" "LIDAL_PLAN_BRANCH_ROWS=1"; do for fr in 5 1; do
  env $cfg timeout 600 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep [$cfg] frames $fr: %.3f ms' % d['ms_per_step'])"
done; done; done
