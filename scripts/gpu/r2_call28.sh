#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c28; mkdir -p $O
timeout 900 python scripts/ablate_wgrad.py shipped 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-variants > $O/bench.log 2>&1; tail -1 $O/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); print({k:(v['ms'],v['calls']) for k,v in d['families'].items()})"
