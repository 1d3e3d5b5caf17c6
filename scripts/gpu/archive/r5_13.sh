#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_13; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_geometry_gpu.py tests/test_data_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/step.log 2>&1; echo "step rc=$?"
cd $GRAFT_REPO_ROOT
for d in score step; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
python3 scripts/gpu/stats_table.py $O/score_kernel_stats.csv 24 60 | grep -E "kmap_probe|table_insert|total|table_query|hash_kernel"
python3 scripts/gpu/stats_table.py $O/step_kernel_stats.csv 10 80 | grep -E "kmap_probe|table_insert|total|table_query|hash_kernel"
