#!/bin/bash
# round 2, GPU call 2: image kernel vs v1 (shapes, orders, tile-shape variants) + the whole GPU test suite
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c2; mkdir -p $O
timeout 600 python scripts/exp_img.py order > $O/exp_img_default.log 2>&1; echo "exp_img rc=$?" | tee -a $O/summary.txt
tail -32 $O/exp_img_default.log
for v in g2w8 g2w4; do
  LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 400 python scripts/exp_img.py > $O/exp_img_$v.log 2>&1; echo "exp_img $v rc=$?" | tee -a $O/summary.txt
  tail -12 $O/exp_img_$v.log
done
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest.log
timeout 900 python -m pytest tests/test_multirank_gpu.py -q > $O/pytest_multirank.log 2>&1; echo "pytest multirank rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_multirank.log
