#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c9; mkdir -p $O
for v in la16 la30 la22; do
EXP_SHAPES=1:96:96,8:256:256,1:32:32 LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1
echo "== $v"; grep "^s" $O/exp_$v.log
done
