#!/bin/bash
O=gpurun_out/r5_cells; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "cells or devoxelize" > $O/tests1.log 2>&1; echo "tests1 rc=$?"; tail -12 $O/tests1.log
timeout 2400 python3 -m pytest tests/test_teacher_forced_gpu.py tests/test_plan_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests2.log 2>&1; echo "tests2 rc=$?"; tail -12 $O/tests2.log
bash scripts/gpu/r5_env_ab.sh LIDAL_DEVOX_CELLS_AVG 12 0
