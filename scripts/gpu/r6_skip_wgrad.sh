#!/bin/bash
# round 6: what ANY speed-up of the level-0 / level-1 weight gradients could give the 5-scan step: the step without them
# (wrong gradients, timing only).  The switch is NOT in the library: the record (profiles/r06_wgrad_streams.txt, 12.78 against
# 13.66 ms) was taken with these two lines at the top of network/plan.py _Run.b_wgrad's launch --
#     if os.environ.get('LIDAL_X_SKIP_WGRAD') == '1' and c.k == 27 and n_x >= 150000:
#         return
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
