#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c12; mkdir -p $O
export EXP_SHAPES=4:128:128,8:256:256,8:384:256,16:256:256,4:64:64
for v in nb4_700 nb4_1700 nb4_4000; do
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1
echo "== $v"; grep "^s" $O/exp_$v.log
done
