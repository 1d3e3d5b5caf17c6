#!/bin/bash
# round 6: steady-state launches per queue with the streamed weight gradients on / off (scripts/gpu/steady_counts.py)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams_steady; mkdir -p $O
B="python3 $GRAFT_REPO_ROOT/bench.py"
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
for rows in 0 150000; do
  cd /tmp
  LIDAL_WGRAD_STREAMS_ROWS=$rows timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$rows -- $B --frames 5 --steps 12 --warmup 3 $QUIET > $O/steady$rows.log 2>&1; echo "steady $rows rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(find $O/t$rows -name "*kernel_trace.csv" | head -1)
  python3 scripts/gpu/steady_counts.py $f 60 > $O/steady_counts_$rows.txt; head -2 $O/steady_counts_$rows.txt; grep "^queue" $O/steady_counts_$rows.txt
  grep "conv_lean_kernelIDF16bLi6ELi192ELi8ELb0\|wgrad_\|onesweep\|rule_key\|scatter_kernel" $O/steady_counts_$rows.txt | cut -c1-120
  rm -rf $O/t$rows
done
