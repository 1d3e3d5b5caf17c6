#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_check2; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
timeout 900 python bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-variants --no-roofline > $O/f32.json 2> $O/f32.err; echo "f32 rc=$?"
python3 - <<PY
import json
d = json.load(open('$O/f32.json'))
print('f32 ms_per_step', d['ms_per_step'])
for k, v in d['families'].items():
    print('  %-20s %s' % (k, json.dumps(v)[:200]))
PY
BENCH_SECONDARY_F32_ONLY=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-families --no-variants > $O/sec.json 2> $O/sec.err; echo "sec rc=$?"
python3 -c "
import json; d=json.load(open('$O/sec.json')); s=d['secondary']; print(json.dumps(s['by_nei']), json.dumps(s['roofline']['scorer']))"
LIDAL_SCORE_CELL_ORDER=0 BENCH_SECONDARY_F32_ONLY=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-families --no-variants > $O/sec0.json 2> $O/sec0.err; echo "sec0 rc=$?"
python3 -c "
import json; d=json.load(open('$O/sec0.json')); s=d['secondary']; print(json.dumps(s['by_nei']), json.dumps(s['roofline']['scorer']))"
