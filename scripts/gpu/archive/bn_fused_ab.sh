#!/bin/bash
# BatchNorm merge kernels inside their consumers (LIDAL_BN_FUSED): unit test, suites that lean on BatchNorm, A/B of the step
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "merges_inside or batch_norm or bn_ or tail" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_plan_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for f in 1 0; do for fr in 1 5; do
  LIDAL_BN_FUSED=$f timeout 600 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep fused $f frames $fr: %.3f ms' % d['ms_per_step'])"
done; done; done
