#!/bin/bash
# round 6: the streamed weight gradients of the large levels ON THE MAIN STREAM (serial, their isolated time) against beside it
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
for rep in 1 2; do
  for cfg in "0 0" "150000 0" "150000 1" "100000 1"; do
    set -- $cfg
    LIDAL_X_STREAMS_MAIN=$2 LIDAL_WGRAD_STREAMS_ROWS=$1 timeout 600 python bench.py --steps 30 --warmup 8 $QUIET > /tmp/b.json 2> /tmp/b.err
    python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('rows>=$1 main $2 rep $rep: ms_per_step', d['ms_per_step'])"
  done
done
