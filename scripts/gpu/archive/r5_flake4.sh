#!/bin/bash
# after the fixes: the bitwise planned-vs-per-operator tests 16 times in a row inside their full file, and the determinism
# script under bf16
fails=0
for r in $(seq 1 8); do
  python3 -m pytest tests/test_plan_gpu.py -x -q -m gpu > /tmp/flake.log 2>&1 || { fails=$((fails+1)); grep "^FAILED" /tmp/flake.log | head -2; }
done
echo "tests/test_plan_gpu.py: $fails failures of 8 runs"
bad=0
for r in 1 2 3; do DTYPE=bf16 REPS=16 python3 scripts/exp/determinism_steps.py 2>&1 | grep -q "runs that differ" && bad=$((bad+1)); done
echo "bf16 determinism: $bad of 3 rounds showed a difference"
