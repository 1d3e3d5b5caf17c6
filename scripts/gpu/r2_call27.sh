#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c27; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $O/ops.log 2>&1; tail -4 $O/ops.log
timeout 900 python scripts/ablate_wgrad.py nodma shipped res1 res3 res4 min8 min32 2>&1 | grep -v amdgpu
