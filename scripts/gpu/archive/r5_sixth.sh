#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_sixth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3300 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"; tail -2 $O/score.log
cd $GRAFT_REPO_ROOT
f=$(find $O/score -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/score_kernel_stats.csv; rm -rf $O/score
python3 scripts/gpu/stats_table.py $O/score_kernel_stats.csv 24 16
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_sixth/bench_line.json')))
print('ms/step', d['ms_per_step'], 'value', d['value'])
print('variants', {k: (v.get('ms_per_step')) for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
print('families', {k: (v.get('ms'), v.get('hbm_frac')) for k, v in d.get('families', {}).items() if isinstance(v, dict)})
print('roofline', d.get('roofline', {}).get('frac'), d.get('roofline', {}).get('launch_us'))
print('secondary', json.dumps(d.get('secondary', {}).get('by_dtype')))
PY
