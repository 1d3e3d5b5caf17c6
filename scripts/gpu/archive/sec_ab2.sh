#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { tag=$1; shift; ( cd $GRAFT_REPO_ROOT; env "$@" timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-families --no-variants --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$tag', {k: x['value'] for k, x in d['secondary']['by_nei'].items()})" ); }
for rep in 1 2 3; do
  run "plan, scoring after (shipped)" X=1
  run "plan, scoring on the table stream" LIDAL_SCORE_OVERLAP=1 LIDAL_SCORE_STREAM=tables
done
