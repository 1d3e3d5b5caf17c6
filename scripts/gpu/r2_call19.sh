#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c19; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-families --no-variants --no-roofline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['ms_per_step'], d['secondary']['by_nei'])
PY
