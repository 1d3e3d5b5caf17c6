#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c21; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad or conv or dense or matmul or linear" > $O/ops.log 2>&1; tail -5 $O/ops.log
timeout 300 python scripts/exp_memorder.py > $O/base.log 2>&1; grep -v amdgpu $O/base.log
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_wgabl1.so timeout 300 python scripts/exp_memorder.py > $O/abl1.log 2>&1; grep -v amdgpu $O/abl1.log
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_nodma.so timeout 300 python scripts/exp_memorder.py > $O/nodma.log 2>&1; grep -v amdgpu $O/nodma.log
