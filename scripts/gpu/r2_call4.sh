#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c4; mkdir -p $O
for v in d1 d2 d3 d3g2w4 d2m3; do
  LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1; echo "$v rc=$?" >> $O/summary.txt
  echo "== $v"; grep "^s\|^dense" $O/exp_$v.log
done
export EXP_SHAPES=1:96:96,8:256:256,1:32:32
for v in d3abl30 d3abl14; do
  LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1; echo "$v rc=$?" >> $O/summary.txt
  echo "== $v"; grep "^s" $O/exp_$v.log
done
