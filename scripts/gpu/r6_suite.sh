#!/bin/bash
# round 6: the full -m gpu suite, then the default bench line (tag = $1)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r6_suite_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-300
python3 - <<PY
import json
d = json.load(open('$O/bench_line.json'))
v = d.get('variants', {})
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in ('single_scan', 'f32', 'f32_exact', 'dropin_surface', 'dropin_surface_no_adopt', 'minkunet', 'fresh_stream', 'inline_geometry', 'score_256'):
    x = v.get(k, {})
    print(k, x.get('ms_per_step', x.get('ms_per_frame')), x.get('error', ''))
s = d.get('secondary', {})
print('secondary', s.get('value'), json.dumps(s.get('by_nei')), json.dumps(s.get('by_dtype', {}).get('f32_exact', {}).get('by_nei')))
print('roofline', json.dumps(d.get('roofline'))[:400])
print('sec roofline', json.dumps(s.get('roofline'))[:900])
PY
