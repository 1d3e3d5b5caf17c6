#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_ninth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_scoring_gpu.py tests/test_data_gpu.py tests/test_benchsize_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "scor or grid or register or adopted or nn_" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/score -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/score_kernel_stats.csv; rm -rf $O/score
python3 scripts/gpu/stats_table.py $O/score_kernel_stats.csv 24 40 | grep -E "interframe|kmap_probe|total|table_"
