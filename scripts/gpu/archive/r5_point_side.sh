#!/bin/bash
O=gpurun_out/r5_point_side; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_plan_gpu.py tests/test_teacher_forced_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
bash scripts/gpu/r5_env_ab.sh LIDAL_PLAN_POINT_SIDE 1 0 2>&1 | grep -v "^$"
