#!/bin/bash
O=gpurun_out/r5_tests_bn; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_plan_gpu.py tests/test_model_gpu.py tests/test_teacher_forced_gpu.py tests/test_robustness_gpu.py tests/test_multirank_gpu.py tests/test_geometry_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
