#!/bin/bash
# how often does the planned-vs-per-operator bitwise test fail, by configuration (12 repetitions each)
for cfg in "default" "LIDAL_DEVOX_CELLS_AVG=0"; do
  fails=0
  for r in $(seq 1 12); do
    if [ "$cfg" = default ]; then python3 -m pytest tests/test_plan_gpu.py -x -q -m gpu -k "bitwise_the_per_operator_steps" > /tmp/flake.log 2>&1 || { fails=$((fails+1)); grep "^FAILED\|At index" /tmp/flake.log | head -2; }
    else env $cfg python3 -m pytest tests/test_plan_gpu.py -x -q -m gpu -k "bitwise_the_per_operator_steps" > /tmp/flake.log 2>&1 || { fails=$((fails+1)); grep "^FAILED\|At index" /tmp/flake.log | head -2; }
    fi
  done
  echo "$cfg: $fails failures of 12"
done
