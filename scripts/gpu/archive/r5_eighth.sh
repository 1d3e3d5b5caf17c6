#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_eighth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "adopted or surface or deferred or prefetched" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-families --no-roofline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_eighth/bench_line.json')))
print('ms/step', d['ms_per_step'])
print('variants', {k: (v.get('ms_per_step'), v.get('library_calls_per_step'), v.get('error')) for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
PY
