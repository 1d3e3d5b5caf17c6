#!/bin/bash
# full GPU test suite + three timed runs of the train step.  usage: check.sh [tag]
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/check_${1:-x}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-families --no-roofline --no-variants 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
