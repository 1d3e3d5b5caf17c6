#!/bin/bash
# rocprofv3 kernel stats of the train step only.  usage: prof_step.sh TAG
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-x}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/step.log 2>&1; echo "step rc=$?"
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob('$O/step/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total GPU ms per step (7 steps profiled):', tot / 1e6 / 7)
for r in rows[:60]:
    print('%-100s %6s %9.1f us  %7.3f ms/step %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / 7, 100 * float(r['TotalDurationNs']) / tot))
PY
tail -1 $O/step.log | cut -c1-200
