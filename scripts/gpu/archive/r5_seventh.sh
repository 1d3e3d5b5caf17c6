#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_seventh; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/test_plan_gpu.py tests/test_robustness_gpu.py tests/test_scoring_gpu.py tests/test_teacher_forced_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
bash scripts/gpu/r5_multirank.sh
cd /tmp
SCORE_DTYPE=f32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score32 -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score32.log 2>&1; echo "score32 rc=$?"; tail -2 $O/score32.log
cd $GRAFT_REPO_ROOT
f=$(find $O/score32 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/score32_kernel_stats.csv; rm -rf $O/score32
python3 scripts/gpu/stats_table.py $O/score32_kernel_stats.csv 24 22
