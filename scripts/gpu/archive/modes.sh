#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do BIND=1 GROUPS=40 timeout 300 python scripts/exp/single_scan_modes.py 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do BIND=1 PRELUDE=5 timeout 300 python scripts/exp/single_scan_modes.py 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do GROUPS=40 timeout 300 python scripts/exp/single_scan_modes.py 2>&1 | grep -v amdgpu.ids; done
