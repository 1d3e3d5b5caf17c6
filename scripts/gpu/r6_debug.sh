#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_debug; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu > $O/model.log 2>&1; echo "model rc=$?"; head -60 $O/model.log | cut -c1-300; echo ...; tail -5 $O/model.log | cut -c1-300
timeout 600 python -m pytest tests/test_scoring_gpu.py -x -q -m gpu -k "256" > $O/s256.log 2>&1; echo "s256 rc=$?"; head -40 $O/s256.log | cut -c1-300; tail -5 $O/s256.log | cut -c1-300
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split" 2>&1 | tail -5
for w in 4 8; do
  echo "=== split32 waves=$w, 8-view frame"
  FRAME=1 LIDAL_SPLIT32_WAVES=$w timeout 600 python scripts/exp/split_check.py 2>&1 | grep -v amdgpu.ids | tee $O/frame_w$w.txt | tail -12
done
echo "=== split32 default, 5-scan batch"
timeout 600 python scripts/exp/split_check.py 2>&1 | grep -v amdgpu.ids | tee $O/batch_default.txt | tail -12
