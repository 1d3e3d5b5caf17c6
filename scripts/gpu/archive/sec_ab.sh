#!/bin/bash
# secondary metric: inference as a plan x scoring beside the inference, same box, the round-3 tree beside it
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { tag=$1; shift; ( cd $GRAFT_REPO_ROOT/$TREE; env "$@" timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-families --no-variants --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s' % '$tag', {k: x['value'] for k, x in d['secondary']['by_nei'].items()})" ); }
for rep in 1 2; do
  TREE=_r3 run "r3" X=1
  TREE=. run "r4 plan, overlap" LIDAL_SCORE_OVERLAP=1
  TREE=. run "r4 plan, no overlap" LIDAL_SCORE_OVERLAP=0
  TREE=. run "r4 per-operator, overlap" LIDAL_PLAN=0 LIDAL_SCORE_OVERLAP=1
  TREE=. run "r4 per-operator, no overlap" LIDAL_PLAN=0 LIDAL_SCORE_OVERLAP=0
  TREE=. run "r4 plan, no overlap, unbound" LIDAL_SCORE_OVERLAP=0 BENCH_NO_AFFINITY=1
  TREE=. run "r4 plan, overlap, unbound" LIDAL_SCORE_OVERLAP=1 BENCH_NO_AFFINITY=1
done
