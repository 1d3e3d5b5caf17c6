#!/bin/bash
# round 6: the rolling-A lean kernel (variant build) against the shipped one, again: layers twice, the bitwise test, the steps
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_roll2; mkdir -p $O
cd $GRAFT_REPO_ROOT
LIDAL_AMD_LIB=scripts/_abl/lib_roll.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bitwise or conv_apply or split_offsets" 2>&1 | tail -3
for rep in 1 2; do
  for lib in "" scripts/_abl/lib_roll.so; do
    SCANS=1 LIDAL_AMD_LIB=$lib timeout 600 python scripts/exp/lean32_check.py 2>&1 | grep "sum of\|384->256\|96->96" | tr '\n' ' '; echo " [$lib]"
  done
done
for lib in "" scripts/_abl/lib_roll.so; do
  for fr in 1 5; do
    LIDAL_AMD_LIB=$lib timeout 600 python bench.py --frames $fr --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('frames $fr lib [$lib]: ms_per_step', d['ms_per_step'])"
  done
done
