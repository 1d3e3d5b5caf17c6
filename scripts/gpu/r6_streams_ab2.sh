#!/bin/bash
# round 6: streamed weight gradients x priority of the weight gradients' queue, 5-scan step
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams_ab2; mkdir -p $O
cd $GRAFT_REPO_ROOT
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
for rep in 1 2; do
  for rows in 0 150000; do
    for pr in 0 -1; do
      LIDAL_X_SIDE_PRIORITY=$pr LIDAL_WGRAD_STREAMS_ROWS=$rows timeout 600 python bench.py --steps 30 --warmup 8 $QUIET > $O/b.json 2> $O/b.err
      python3 -c "
import json; d=json.load(open('$O/b.json')); print('rows>=$rows priority $pr rep $rep: ms_per_step', d['ms_per_step'])"
    done
  done
done
