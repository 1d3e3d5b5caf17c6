#!/bin/bash
# round 3: full GPU suite, default bench line, kernel stats of the train step.  usage: r3_check.sh TAG
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3_${1:-x}; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests.log 2>&1; tail -30 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python3 - <<PY
import json
try:
    d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
    print('ms/step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['launch_us'])
    print({k: (v.get('ms'), v.get('hbm_frac')) for k, v in d.get('families', {}).items()})
    print(d.get('variants')); print(d.get('secondary', {}).get('by_nei'))
except Exception as e:
    print('bench parse failed', e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-secondary --no-families --no-roofline --no-variants > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/step_kernel_stats.csv
rm -rf $O/prof
python3 scripts/gpu/stats_table.py $O/step_kernel_stats.csv 10 | head -70
