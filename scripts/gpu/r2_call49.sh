#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "radix_sort" 2>&1 | tail -5
timeout 300 python scripts/sort_bench.py 2>&1 | grep -v amdgpu
