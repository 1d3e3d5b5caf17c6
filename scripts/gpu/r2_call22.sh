#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LIDAL_EXP_ORDERS=dataset
O=gpurun_out/r2c22; mkdir -p $O
for v in abl1 abl2 abl4 abl5 d3 d1; do echo "== $v"; LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_memorder.py 2>&1 | grep -v amdgpu; done
for c in 1024 2048 3072; do echo "== chunk $c"; LIDAL_EXP_CHUNK=$c timeout 300 python scripts/exp_memorder.py 2>&1 | grep -v amdgpu; done
