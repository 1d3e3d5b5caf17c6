#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -25 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['frac'])
print({k:v.get('ms') for k,v in d['families'].items()})
print(d['variants'])
PY
