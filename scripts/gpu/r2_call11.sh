#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c11; mkdir -p $O
timeout 400 python scripts/exp_img.py > $O/exp_lean.log 2>&1; echo "lean rc=$?" >> $O/summary.txt
grep "^s\|^dense" $O/exp_lean.log
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_multirank_gpu.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
grep "gradfull\|^spvcnn \[\|^minkunet \[\|bf16 argmax" $O/pytest.log | head; tail -6 $O/pytest.log
BENCH_VERBOSE=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
tail -3 $O/bench.err; cat $O/bench.json
