#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c41; mkdir -p $O
for w in 0 1; do LIDAL_WGRAD_STREAM=$w timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-families --no-roofline > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('stream=$w', d['ms_per_step'], d['value'], {k:v.get('ms_per_step') for k,v in d.get('variants',{}).items()})"; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
