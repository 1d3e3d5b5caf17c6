#!/bin/bash
# prefetcher: thread on / off x run-ahead bound 1 / 2 / 3: single scan, 5 scans, secondary; same box
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "0 1" "0 2" "1 2" "1 3" "0 3"; do
  set -- $cfg
  for fr in 1 5; do
  LIDAL_GEOMETRY_THREAD=$1 LIDAL_GEOMETRY_PENDING=$2 timeout 900 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants $([ $fr = 1 ] && echo --no-secondary) 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep thread $1 pending $2 frames $fr: %.3f ms' % d['ms_per_step'], {k: x['value'] for k, x in d.get('secondary', {}).get('by_nei', {}).items()})"
  done
done; done
