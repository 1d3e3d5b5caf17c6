#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c15; mkdir -p $O
export EXP_SHAPES=1:96:96,8:256:256,1:32:32,4:128:128
for v in la32 la38 la6; do
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1
echo "== $v"; grep "^s" $O/exp_$v.log
done
