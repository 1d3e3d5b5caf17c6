#!/bin/bash
O=gpurun_out/r5_tail_rows; mkdir -p $O
for v in 100000 1 30000 100000 1 30000; do
  export LIDAL_TAIL_SUMS_ROWS=$v
  for f in 5 1; do
  timeout 600 python3 bench.py --frames $f --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line_$v.json 2> $O/err_$v.txt
  python3 -c "
import json
d=json.load(open('$O/line_$v.json'))
print('rows $v frames $f: step', d['ms_per_step'], 'inline', d['families']['whole_step']['ms'], 'bn', d['families']['batch_norm']['ms'], 'ew', d['families']['fused_elementwise']['ms'])
"
  done
done
