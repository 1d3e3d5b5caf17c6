#!/bin/bash
# round 6: per-kernel times of lidal_wgrad_streams_build and the streamed weight gradient (scripts/exp/wgrad_streams.py)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams_prof; mkdir -p $O
cd /tmp
LEVEL=${LEVEL:-0} BLOCK=1024 W=512 REPS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $GRAFT_REPO_ROOT/scripts/exp/wgrad_streams.py > $O/log.txt 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/st -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/st
python3 scripts/gpu/stats_table.py $O/kernel_stats.csv 1 40
