#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c16; mkdir -p $O
timeout 400 python scripts/exp_img.py > $O/exp.log 2>&1; grep "^s\|^dense" $O/exp.log
ABL_DTYPE=f32 timeout 400 python scripts/exp_img.py > $O/exp_f32.log 2>&1; grep "^s\|^dense" $O/exp_f32.log
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-variants > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['frac'])
print({k:v.get('ms') for k,v in d['families'].items()})
PY
