#!/bin/bash
# main-queue gaps of the steady-state step (kernel trace of bench.py, scripts/gpu/steady_counts.py)
O=gpurun_out/r5_gaps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t5 -- python3 $GRAFT_REPO_ROOT/bench.py --frames 5 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $GRAFT_REPO_ROOT/$O/steady5.log 2>&1; echo "steady rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/t5 -name '*kernel_trace.csv' | head -1)
python3 scripts/gpu/steady_counts.py $f 8 > $O/steady_counts_gaps.txt
grep -n "^gap of" -A 60 $O/steady_counts_gaps.txt | head -150
rm -rf $O/t5
