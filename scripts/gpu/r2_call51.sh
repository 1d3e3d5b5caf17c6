#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c51; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['variants'].items()}, d['secondary']['by_nei'])"
