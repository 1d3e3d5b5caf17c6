#!/bin/bash
O=gpurun_out/r5_slab; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_model_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log
for v in 1 0 1 0; do
  export LIDAL_BN_SLAB_SUMS=$v
  for f in 5 1; do
  timeout 600 python3 bench.py --frames $f --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line_$v.json 2> $O/err_$v.txt
  python3 -c "
import json
d=json.load(open('$O/line_$v.json'))
print('slab $v frames $f: step', d['ms_per_step'], 'inline', d['families']['whole_step']['ms'], 'bn', d['families']['batch_norm']['ms'], 'ew', d['families']['fused_elementwise']['ms'])
"
  done
done
