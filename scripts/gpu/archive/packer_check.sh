#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/packer; mkdir -p $O
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_multirank_gpu.py tests/test_robustness_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -5
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/bench.py --frames 1 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/p.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/p -name '*kernel_stats.csv' | head -1); grep "weight_image" $f | cut -c1-200; rm -rf $O/p
for fr in 1 5; do timeout 600 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames $fr: %.3f ms' % d['ms_per_step'])"; done
