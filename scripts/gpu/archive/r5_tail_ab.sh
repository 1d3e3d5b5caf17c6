#!/bin/bash
# the fused tail of a residual block's backward on every level (LIDAL_TAIL_SUMS_ROWS) against the default row limit
O=gpurun_out/r5_tail_ab; mkdir -p $O
for v in default all default all; do
  if [ $v = all ]; then export LIDAL_TAIL_SUMS_ROWS=100000000; else unset LIDAL_TAIL_SUMS_ROWS; fi
  BENCH_FAMILY_CALLS=$O/calls_$v.jsonl timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-secondary > $O/line_$v.json 2> $O/err_$v.txt
  python3 - $v <<'PY'
import json, sys, collections
v=sys.argv[1]
d=json.load(open('gpurun_out/r5_tail_ab/line_%s.json'%v))
rows=[json.loads(l) for l in open('gpurun_out/r5_tail_ab/calls_%s.jsonl'%v)]
t=collections.OrderedDict()
for r in rows:
    if r['family'] not in ('batch_norm','fused_elementwise'): continue
    d2=t.setdefault(r['name'],[0,0.0]); d2[0]+=1; d2[1]+=r['ms']
print(v, 'step', d['ms_per_step'], 'bn', d['families']['batch_norm']['ms'], 'ew', d['families']['fused_elementwise']['ms'], ' '.join('%s %.3f'%(k.replace('lidal_',''),x[1]) for k,x in t.items()))
if v=='all':
    for r in rows:
        if r['name'] in ('lidal_add_relu_bwd_bn_sums',): print('   ', r['name'], r['ms'], [a for a in r['args'] if 0<a<10**7][:4])
PY
done
