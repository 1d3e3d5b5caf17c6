#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py -m gpu -q -x 2>&1 | tail -2
bash scripts/gpu/steady.sh pl1 1 2>&1 | grep -i "bn_apply_tiles\|bn_bwd_dx_merge\|steady state\|^queue 1"
bash scripts/gpu/steady.sh pl5 5 2>&1 | grep -i "bn_apply_tiles\|bn_bwd_dx_merge\|steady state\|^queue 1"
for fr in 1 5 1 5; do timeout 600 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $fr: %.3f ms' % d['ms_per_step'])"; done
