#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/exp/ddp_probe.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM\|Gloo" | tail -${TAIL:-8}; }
TAIL=50 run PROFILE=1 FRAMES=1
run LIDAL_PLAN_SIDE_ROWS=0 LIDAL_PLAN_BRANCH_ROWS=1000000000 FRAMES=1
run LIDAL_PLAN_BLOCK_MB=64 FRAMES=1
echo "== score timeline"
O=$GRAFT_REPO_ROOT/gpurun_out/score_tl; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/scripts/exp/score_timeline.py 32 10 > $O/run.log 2>&1; tail -3 $O/run.log
cd $GRAFT_REPO_ROOT
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 scripts/exp/score_timeline_read.py $f 32 > $O/timeline.txt; cat $O/timeline.txt
rm -rf $O/t
