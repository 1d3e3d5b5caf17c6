#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in noskip abl2 abl4 abl8; do echo "== $v"; LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py 2>&1 | grep -v amdgpu | grep "^s4  128\|^s8\|^s16\|s1   96"; done
