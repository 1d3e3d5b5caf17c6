#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_tenth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_teacher_forced_gpu.py tests/test_plan_gpu.py -q -m gpu -x -s -k "inference" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "inference|ops |passed|failed|Error" $O/tests.log | head -40
