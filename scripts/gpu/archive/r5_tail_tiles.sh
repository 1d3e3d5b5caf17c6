#!/bin/bash
O=gpurun_out/r5_tail_tiles; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "tail or merges_inside" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
for v in on off on off; do
  if [ $v = off ]; then export LIDAL_TAIL_TILES=0; else unset LIDAL_TAIL_TILES; fi
  BENCH_FAMILY_CALLS=$O/calls_$v.jsonl timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-secondary --no-roofline > $O/line_$v.json 2> $O/err_$v.txt
  python3 - $v <<'PY'
import json, sys, collections
v=sys.argv[1]
d=json.load(open('gpurun_out/r5_tail_tiles/line_%s.json'%v))
rows=[json.loads(l) for l in open('gpurun_out/r5_tail_tiles/calls_%s.jsonl'%v)]
t=collections.OrderedDict()
for r in rows:
    if r['family'] not in ('batch_norm','fused_elementwise','other_lib'): continue
    d2=t.setdefault(r['name'],[0,0.0]); d2[0]+=1; d2[1]+=r['ms']
print(v, 'step', d['ms_per_step'], 'inline', d['families']['whole_step']['ms'], 'bn', d['families']['batch_norm']['ms'], 'ew', d['families']['fused_elementwise']['ms'], ' '.join('%s %.3f'%(k.replace('lidal_',''),x[1]) for k,x in t.items()))
PY
done
