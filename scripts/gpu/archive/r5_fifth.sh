#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_fifth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/exp/split_check.py 2>&1 | tail -30
timeout 1200 python -m pytest tests -q -m gpu -x -k "surface or deferred or scoring or prob_inference or planned_inference or plain_torchsparse" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dropin -- python3 $GRAFT_REPO_ROOT/scripts/exp/dropin_profile.py 8 > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -2 $O/dropin.log
cd $GRAFT_REPO_ROOT
f=$(find $O/dropin -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/dropin_kernel_stats.csv; rm -rf $O/dropin
python3 scripts/gpu/stats_table.py $O/dropin_kernel_stats.csv 10 40
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-families --no-variants --no-roofline --score-frames 16 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_fifth/bench_line.json')))
print('ms/step', d['ms_per_step'])
print('secondary', json.dumps(d.get('secondary', {}).get('by_dtype')))
PY
