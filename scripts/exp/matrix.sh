#!/bin/bash
# step time over {offset split} x {shortcut branch stream threshold} x {weight-gradient stream}, 1 and 5 scans, 2 rounds
for rep in 1 2; do for fr in 1 5; do for cfg in "1 30000 1099511627776" "0 30000 1099511627776" "1 0 1099511627776" "0 0 1099511627776" "0 0 0" "1 0 0"; do
  set -- $cfg
  LIDAL_CONV_SPLIT=$1 LIDAL_PLAN_BRANCH_ROWS=$2 LIDAL_PLAN_SIDE_ROWS=$3 python bench.py --frames $fr --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep frames $fr split $1 branch_rows $2 side_rows $3 ms/step', d['ms_per_step'])"
done; done; done
