#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/exp_bn.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad or dense or conv" 2>&1 | tail -2
timeout 300 python scripts/ablate_wgrad.py shipped 2>&1 | grep -v amdgpu
