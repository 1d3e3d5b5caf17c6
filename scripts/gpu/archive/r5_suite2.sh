#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_suite; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3300 python -m pytest tests/test_model_gpu.py tests/test_multirank_gpu.py tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_robustness_gpu.py tests/test_scoring_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests2.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests2.log
