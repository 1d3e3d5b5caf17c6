#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c20; mkdir -p $O
timeout 300 python scripts/exp_memorder.py > $O/base.log 2>&1; grep -v amdgpu $O/base.log
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_wgabl1.so timeout 300 python scripts/exp_memorder.py > $O/abl1.log 2>&1; grep -v amdgpu $O/abl1.log
