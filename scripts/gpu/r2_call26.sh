#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c26; mkdir -p $O
LIDAL_ABL_CHUNKS=1536,2048,2560,3072,4096,6144,8192 timeout 900 python scripts/ablate_wgrad.py shipped 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-variants > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1500
