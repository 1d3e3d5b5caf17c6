#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROFILE_STACKS=1
timeout 600 python scripts/profile_ops.py 2 2>&1 | grep -v amdgpu | grep "/step" | sort -k2 -n -r | head -60
