#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_16; mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/exp/bn_merge_cost.py 2>&1 | tail -9
timeout 3000 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_plan_gpu.py tests/test_teacher_forced_gpu.py tests/test_benchsize_gpu.py tests/test_robustness_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_16/bench_line.json')))
print('ms/step', d['ms_per_step'])
print('families', {k: v.get('ms') for k, v in d['families'].items()})
print('variants', {k: v.get('ms_per_step') for k, v in d['variants'].items() if isinstance(v, dict)})
PY
