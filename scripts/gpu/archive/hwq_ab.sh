#!/bin/bash
# hardware queues per process (GPU_MAX_HW_QUEUES; ROCm's default is 4) against the five streams of a step / a scored frame
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for q in 4 8 6; do
  GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['variants']
print('rep $rep queues $q: 5 scans %.3f  inline %.3f  single %.3f  fresh_stream %.3f  minkunet %.3f  f32 %.3f | frames/s' % (d['ms_per_step'], v['inline_geometry']['ms_per_step'], v['single_scan']['ms_per_step'], v['fresh_stream']['ms_per_step'], v['minkunet']['ms_per_step'], v['f32']['ms_per_step']), {k: x['value'] for k, x in d['secondary']['by_nei'].items()})"
done; done
