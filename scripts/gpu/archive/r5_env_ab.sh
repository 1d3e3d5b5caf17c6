#!/bin/bash
# A/B of one environment variable on the headline step and the single scan: bash scripts/gpu/r5_env_ab.sh NAME v1 v2 ...
O=gpurun_out/r5_env_ab; mkdir -p $O
name=$1; shift
for v in "$@" "$@"; do
  export $name=$v
  for f in 5 1; do
  timeout 600 python3 bench.py --frames $f --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line.json 2> $O/err.txt
  python3 -c "
import json
d=json.load(open('$O/line.json'))
fm=d['families']
print('$name=$v frames $f: step', d['ms_per_step'], 'inline', fm['whole_step']['ms'], 'conv', fm['conv_apply']['ms'], 'bn', fm['batch_norm']['ms'], 'wgrad', fm['conv_wgrad']['ms'], 'pv', fm['point_voxel']['ms'])
"
  done
done
