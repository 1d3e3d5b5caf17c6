#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c8; mkdir -p $O
timeout 400 python scripts/exp_img.py > $O/exp_lean.log 2>&1; echo "lean rc=$?" >> $O/summary.txt
grep "^s\|^dense" $O/exp_lean.log
ABL_DTYPE=f32 timeout 400 python scripts/exp_img.py > $O/exp_lean_f32.log 2>&1; echo "lean f32 rc=$?" >> $O/summary.txt
grep "^s\|^dense" $O/exp_lean_f32.log
for v in la6 la14 wg1; do
EXP_SHAPES=1:96:96,8:256:256,1:32:32 LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py > $O/exp_$v.log 2>&1
echo "== $v"; grep "^s" $O/exp_$v.log
done
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -8 $O/pytest.log
