#!/bin/bash
# weight gradients on the side stream only for layers of at least MIN rows; shortcut branch threshold BR
for rep in 1 2 3; do for fr in 1 5; do for cfg in "0 0" "40000 0" "60000 0" "100000 0" "60000 60000" "100000 100000"; do
  set -- $cfg
  LIDAL_PLAN_SIDE_MIN_ROWS=$1 LIDAL_PLAN_BRANCH_ROWS=$2 python bench.py --frames $fr --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep frames $fr side_min_rows $1 branch_rows $2 ms/step', d['ms_per_step'])"
done; done; done
