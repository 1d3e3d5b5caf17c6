#!/bin/bash
# round 5, second GPU call: phase stamps of the lean kernel, f32 scoring kernel stats, the failed test again
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_second; mkdir -p $O
cd $GRAFT_REPO_ROOT
LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_stamps.so timeout 600 python scripts/exp/phase_stamps.py $O/phase_stamps.json > $O/stamps.log 2>&1; echo "stamps rc=$?"; cat $O/stamps.log | tail -60
timeout 900 python -m pytest tests -q -m gpu -x -k "starts_its_own_ranks or one_weight_only or two_rank_ddp" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
cd /tmp
SCORE_DTYPE=f32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score32 -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score32.log 2>&1; echo "score32 rc=$?"; tail -3 $O/score32.log
cd $GRAFT_REPO_ROOT
f=$(find $O/score32 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/score32_kernel_stats.csv; rm -rf $O/score32
python3 scripts/gpu/stats_table.py $O/score32_kernel_stats.csv 24 25
