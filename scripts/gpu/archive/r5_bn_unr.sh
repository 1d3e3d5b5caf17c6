#!/bin/bash
# BatchNorm family of one profiled step under A/B builds of bn.hip (scripts/_abl/lib_<name>.so; `stock` = the product library)
#   bash scripts/gpu/r5_bn_unr.sh stock bn_nopipe stock bn_nopipe
O=gpurun_out/r5_bn_unr; mkdir -p $O
for v in ${@:-stock bn_a8 bn_p8 bn_d8 bn_all8 stock}; do
  if [ $v = stock ]; then unset LIDAL_AMD_LIB; else export LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so; fi
  BENCH_FAMILY_CALLS=$O/calls_$v.jsonl timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-secondary > $O/line_$v.json 2> $O/err_$v.txt
  python3 - $v <<'PY'
import json, sys, collections
v=sys.argv[1]
d=json.load(open('gpurun_out/r5_bn_unr/line_%s.json'%v))
rows=[json.loads(l) for l in open('gpurun_out/r5_bn_unr/calls_%s.jsonl'%v)]
t=collections.OrderedDict()
for r in rows:
    if r['family']!='batch_norm': continue
    d2=t.setdefault(r['name'],[0,0.0]); d2[0]+=1; d2[1]+=r['ms']
big=[r for r in rows if r['name']=='lidal_bn_bwd' and 396662 in r['args']]
print('   bn_bwd at 396662 rows:', ' '.join('%d:%.1f'%(r['args'][-1] if False else [a for a in r['args'] if 0<a<1000][-1], r['ms']*1e3) for r in big))
print(v, 'step', d['ms_per_step'], 'bn', d['families']['batch_norm']['ms'], ' '.join('%s %.3f'%(k.replace('lidal_',''),x[1]) for k,x in t.items()))
PY
done
