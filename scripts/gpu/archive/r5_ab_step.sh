#!/bin/bash
# A/B of library builds on the headline step and the single scan: bash scripts/gpu/r5_ab_step.sh stock sp300 stock sp300
O=gpurun_out/r5_ab_step; mkdir -p $O
for v in "$@"; do
  if [ $v = stock ]; then unset LIDAL_AMD_LIB; else export LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so; fi
  for f in 5 1; do
    BENCH_FAMILY_CALLS=$O/calls_${v}_$f.jsonl timeout 600 python3 bench.py --frames $f --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line_${v}_$f.json 2> $O/err_${v}_$f.txt
    python3 -c "
import json
d=json.load(open('$O/line_${v}_$f.json'))
print('$v frames $f: step', d['ms_per_step'], 'conv_apply', d['families']['conv_apply']['ms'], 'inline', d['families']['whole_step']['ms'])
"
  done
done
