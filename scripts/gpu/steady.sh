#!/bin/bash
# steady-state launches per step by queue.  usage: steady.sh TAG FRAMES
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-x}; FR=${2:-1}
O=$GRAFT_REPO_ROOT/gpurun_out/steady_$TAG; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/bench.py --frames $FR --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 scripts/gpu/steady_counts.py $f 40 > $O/steady_counts.txt; head -100 $O/steady_counts.txt
rm -f $f
