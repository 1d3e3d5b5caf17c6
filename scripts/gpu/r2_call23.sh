#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LIDAL_EXP_ORDERS=dataset
for v in abl2 abl6 abl10 abl18; do echo "== $v"; LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_memorder.py 2>&1 | grep -v amdgpu; done
