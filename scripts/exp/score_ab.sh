#!/bin/bash
# secondary metric: scoring beside inference (third stream) x shortcut on a side stream in the inference plan
for rep in 1 2; do for cfg in "1 100000" "0 100000" "1 0" "0 0"; do
  set -- $cfg
  LIDAL_SCORE_OVERLAP=$1 LIDAL_PLAN_BRANCH_ROWS=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep overlap $1 branch_rows $2', json.dumps(d['secondary']['by_nei']))"
done; done
