#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/exp_bn.py 2>&1 | grep -v amdgpu
