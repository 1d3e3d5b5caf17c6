#!/bin/bash
# the prefetcher's thread on / off: tests, then the bench line's step times and the secondary metric, same box
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/thread_ab; mkdir -p $O
timeout 1500 python -m pytest tests/test_geometry_gpu.py tests/test_plan_gpu.py tests/test_scoring_gpu.py tests/test_data_gpu.py -m gpu -q -x 2>&1 | tail -15
for rep in 1 2; do for th in 1 0; do
  LIDAL_GEOMETRY_THREAD=$th timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families > $O/line_${th}_$rep.json 2> $O/err_${th}_$rep.log
  python - $O/line_${th}_$rep.json $th $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v = d['variants']
print('thread %s rep %s: 5 scans %.3f  inline %.3f  single %.3f  fresh_coords %.3f  fresh_stream %.3f  minkunet %.3f  | frames/s %s' % (
    sys.argv[2], sys.argv[3], d['ms_per_step'], v['inline_geometry']['ms_per_step'], v['single_scan']['ms_per_step'], v['fresh_coords']['ms_per_step'],
    v['fresh_stream']['ms_per_step'], v['minkunet']['ms_per_step'], {k: x['value'] for k, x in d['secondary']['by_nei'].items()}), d.get('host'))
PY
done; done
