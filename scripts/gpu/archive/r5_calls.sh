#!/bin/bash
# per-call table of one profiled step (bench.py family_table with BENCH_FAMILY_CALLS)
O=gpurun_out/r5_calls; mkdir -p $O
BENCH_FAMILY_CALLS=$O/calls.jsonl timeout 900 python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-variants --no-secondary > $O/bench_line.json 2> $O/bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r5_calls/calls.jsonl')]
tot=collections.Counter()
for r in rows: tot[r['family']]+=r['ms']
print('families', {k: round(v,3) for k,v in tot.most_common()})
for fam in ('batch_norm','point_voxel','kernel_maps','fused_elementwise','weight_pack','other_lib'):
    t=collections.OrderedDict()
    for r in rows:
        if r['family']!=fam: continue
        k=(r['name'],tuple(v for v in r['args'] if 0<v<10**7)[:4])
        d=t.setdefault(k,[0,0.0]); d[0]+=1; d[1]+=r['ms']
    print(fam, round(sum(d[1] for d in t.values()),3))
    for k,d in sorted(t.items(), key=lambda kv:-kv[1][1])[:40]:
        print('  %7.1f us x %2d  %s %s'%(d[1]/d[0]*1e3,d[0],k[0],k[1]))
PY
