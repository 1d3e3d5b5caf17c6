#!/bin/bash
O=gpurun_out/r5_img; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_model_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for f in 5 1; do
  timeout 600 python3 bench.py --frames $f --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line.json 2> $O/err.txt
  python3 -c "
import json
d=json.load(open('$O/line.json'))
fm=d['families']
print('frames $f: step', d['ms_per_step'], 'inline', fm['whole_step']['ms'], 'weight_pack', fm['weight_pack']['ms'], 'conv', fm['conv_apply']['ms'])
"
done
