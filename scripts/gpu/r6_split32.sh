#!/bin/bash
# round 6: the 32 x 32 x 16 form of the f32 split convolution (conv_split32_kernel) against the first form -- tests, per-layer
# A/B on an 8-view frame and on the 5-scan batch, f32 frames/s with either form
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_split32; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split" 2>&1 | tail -5
for form in 16 32; do
  echo "=== LIDAL_SPLIT_MFMA=$form, 8-view frame"
  FRAME=1 LIDAL_SPLIT_MFMA=$form timeout 600 python scripts/exp/split_check.py 2>&1 | tee $O/frame_$form.txt | tail -12
  echo "=== LIDAL_SPLIT_MFMA=$form, 5-scan batch"
  LIDAL_SPLIT_MFMA=$form timeout 600 python scripts/exp/split_check.py 2>&1 | tee $O/batch_$form.txt | tail -12
done
for form in 16 32; do
  echo "=== LIDAL_SPLIT_MFMA=$form, f32 frames/s"
  BENCH_SECONDARY_F32_ONLY=1 LIDAL_SPLIT_MFMA=$form timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-families --no-variants > $O/bench_$form.json 2> $O/bench_$form.err
  python3 -c "import json,sys; d=json.load(open('$O/bench_$form.json')); print(d['ms_per_step'], json.dumps(d['secondary']['by_nei']))"
done
timeout 1200 python -m pytest tests/test_teacher_forced_gpu.py tests/test_scoring_gpu.py tests/test_plan_gpu.py -x -q -m gpu 2>&1 | tail -5
