#!/bin/bash
# the default bench line with / without the CPU binding (BENCH_NO_AFFINITY), full defaults (CPU baselines first)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/bind; mkdir -p $O
for rep in 1 2 3; do for na in "" 1; do
  BENCH_VERBOSE=1 BENCH_NO_AFFINITY=$na timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_${na:-0}_$rep.json 2> $O/err_${na:-0}_$rep.log
  grep "bound to" $O/err_${na:-0}_$rep.log | tail -1
  python - $O/line_${na:-0}_$rep.json "${na:-0}" $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); v = d['variants']
print('no_affinity %s rep %s: 5 scans %.3f  inline %.3f  single %.3f  fresh_stream %.3f  minkunet %.3f  f32 %.3f | frames/s %s | roof %.1f us' % (
    sys.argv[2], sys.argv[3], d['ms_per_step'], v['inline_geometry']['ms_per_step'], v['single_scan']['ms_per_step'], v['fresh_stream']['ms_per_step'],
    v['minkunet']['ms_per_step'], v['f32']['ms_per_step'], {k: x['value'] for k, x in d['secondary']['by_nei'].items()}, d['roofline']['launch_us']))
PY
done; done
