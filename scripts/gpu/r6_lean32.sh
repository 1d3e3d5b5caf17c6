#!/bin/bash
# round 6: conv_lean32_kernel (32 x 32 x 16 MFMA, 256-row tiles, one rolling set of A fragments) against the lean kernel
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_lean32; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  LIDAL_LEAN32=$v timeout 600 python scripts/exp/lean32_check.py 2>&1 | grep -v amdgpu.ids | tee $O/layers_$v.txt
  LIDAL_LEAN32=$v timeout 600 python bench.py --roofline-only 2>/dev/null | tee $O/roof_$v.json
  SCANS=1 LIDAL_LEAN32=$v timeout 600 python scripts/exp/lean32_check.py 2>&1 | grep -v amdgpu.ids | tee $O/layers1_$v.txt
done
timeout 600 python -m pytest tests/test_scoring_gpu.py -x -q -m gpu -k "256" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5
