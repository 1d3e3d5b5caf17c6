#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_pair; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/exp/pair_bn_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/pair.txt
LIDAL_BN_FUSED=0 timeout 900 python scripts/exp/pair_bn_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/pair_unfused.txt
