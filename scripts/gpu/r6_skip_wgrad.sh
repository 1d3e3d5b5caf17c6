#!/bin/bash
# round 6: what ANY speed-up of the level-0 / level-1 weight gradients could give the 5-scan step: the step without them
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
for rep in 1 2; do
  for skip in 0 1; do
    LIDAL_X_SKIP_WGRAD=$skip LIDAL_WGRAD_STREAMS_ROWS=0 timeout 600 python bench.py --steps 30 --warmup 8 $QUIET > /tmp/b.json 2> /tmp/b.err
    python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('skip $skip rep $rep: ms_per_step', d['ms_per_step'])"
  done
done
