#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_twelfth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_teacher_forced_gpu.py tests/test_plan_gpu.py tests/test_scoring_gpu.py tests/test_ops_gpu.py -q -m gpu -x -k "inference or scor or split or prob" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-families --no-variants --no-roofline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_twelfth/bench_line.json')))
print('ms/step', d['ms_per_step'])
print('secondary', json.dumps({k: v['by_nei'] for k, v in d['secondary']['by_dtype'].items()}))
PY
