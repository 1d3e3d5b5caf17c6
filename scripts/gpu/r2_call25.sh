#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/ablate_wgrad.py nodma shipped d1 d2 2>&1 | grep -v amdgpu
