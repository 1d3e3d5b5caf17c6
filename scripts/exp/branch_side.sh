#!/bin/bash
# the shortcut branch of the residual blocks on a third stream: step time (1 and 5 scans), 3 rounds
for rep in 1 2 3; do for fr in 1 5; do for b in 0 1 30000; do
  LIDAL_PLAN_BRANCH_ROWS=$b python bench.py --frames $fr --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep frames $fr branch_side $b ms/step', d['ms_per_step'])"
done; done; done
