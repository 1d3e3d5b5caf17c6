#!/bin/bash
# round 6: the lean kernel with ONE rolling set of A fragments and one loop body (-DLIDAL_LEAN_ROLLING) against the shipped one
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_roll; mkdir -p $O
cd $GRAFT_REPO_ROOT
for lib in "" scripts/_abl/lib_roll.so; do
  LIDAL_AMD_LIB=$lib timeout 600 python scripts/exp/lean32_check.py 2>&1 | grep -v amdgpu.ids | tee $O/layers_$(basename "$lib" .so).txt
  SCANS=1 LIDAL_AMD_LIB=$lib timeout 600 python scripts/exp/lean32_check.py 2>&1 | grep -v amdgpu.ids | tee $O/layers1_$(basename "$lib" .so).txt
done
timeout 600 python -m pytest tests/test_scoring_gpu.py -x -q -m gpu -k "256" 2>&1 | tail -5
