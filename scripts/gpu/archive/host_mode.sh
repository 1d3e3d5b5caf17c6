#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 300 python scripts/exp/host_mode_probe.py 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== local affinity"
for i in 1 2 3 4; do AFFINITY=local timeout 300 python scripts/exp/host_mode_probe.py 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== node 0 / node 1"
for i in 1 2; do AFFINITY=0-63,128-191 timeout 300 python scripts/exp/host_mode_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; AFFINITY=64-127,192-255 timeout 300 python scripts/exp/host_mode_probe.py 2>&1 | grep -v amdgpu.ids | tail -1; done
