#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c13; mkdir -p $O
timeout 300 python scripts/exp_bn.py > $O/bn_256.log 2>&1; cat $O/bn_256.log | grep -v amdgpu
for v in bn512 bn1024 bn2048; do
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_bn.py > $O/$v.log 2>&1; cat $O/$v.log | grep -v amdgpu
done
