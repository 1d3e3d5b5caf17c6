#!/bin/bash
# weight gradients on a second stream inside the backward plan: step time by row threshold (1 and 5 scans), 3 rounds
for rep in 1 2 3; do for fr in 1 5; do for r in ${ROWS:-0 1000000}; do
  LIDAL_PLAN_SIDE_ROWS=$r python bench.py --frames $fr --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep frames $fr side_rows $r ms/step', d['ms_per_step'])"
done; done; done
