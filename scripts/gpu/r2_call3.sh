#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c3; mkdir -p $O
export EXP_SHAPES=1:96:96,8:256:256,1:32:32
for v in abl1 abl2 abl4 abl8 abl16 abl12 abl14 abl30; do
  LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1; echo "$v rc=$?" >> $O/summary.txt
  echo "== $v"; grep "^s" $O/exp_$v.log
done
unset EXP_SHAPES
timeout 300 python scripts/exp_memorder.py > $O/memorder.log 2>&1; echo "memorder rc=$?" >> $O/summary.txt; tail -5 $O/memorder.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -8 $O/pytest.log
