#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/profile_ops.py 3 2>&1 | grep -v amdgpu | tail -75
