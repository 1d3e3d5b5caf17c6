#!/bin/bash
# flake rate after moving the f32 weight gradients to the main stream (5 rounds of 16 repetitions), and what f32 costs now
for cfg in "A=1" "LIDAL_PLAN=0" "LIDAL_PLAN_SIDE_F32=1"; do
  bad=0
  for r in 1 2 3 4 5; do
    env $cfg REPS=16 python3 scripts/exp/determinism_steps.py 2>&1 | grep -q "runs that differ" && bad=$((bad+1))
  done
  echo "$cfg: $bad of 5 rounds showed a difference"
done
for cfg in "A=1" "LIDAL_PLAN_SIDE_F32=1"; do
  env $cfg python3 bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-secondary --no-families --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg f32 step', d['ms_per_step'])"
done
