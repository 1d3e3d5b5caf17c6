#!/bin/bash
# the round-3 tree (_r3: git worktree of f77168e, built in place) against this tree, same box, same flow
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r3r4; mkdir -p $O
for rep in 1 2; do for t in _r3 .; do
  cd $GRAFT_REPO_ROOT/$t
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-families > $O/line_${t//[._]/x}_$rep.json 2>/dev/null
  python - $O/line_${t//[._]/x}_$rep.json "$t" $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); v = d.get('variants', {})
print('tree %-4s rep %s: 5 scans %.3f | single %.3f | fresh_coords %.3f | minkunet %.3f | frames/s %s | roof %.1f us' % (
    sys.argv[2], sys.argv[3], d['ms_per_step'], v.get('single_scan', {}).get('ms_per_step', 0), v.get('fresh_coords', {}).get('ms_per_step', 0),
    v.get('minkunet', {}).get('ms_per_step', 0), {k: x['value'] for k, x in d['secondary']['by_nei'].items()}, d['roofline']['launch_us']))
PY
done; done
