#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in bn_u8_w256 bn_u4_w512 bn_u8_w512 bn_u4_w1024 bn_u8_w1024 bn_u4_w2048 bn_u8_w2048; do echo "== $v"; LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_bn.py 2>&1 | grep -v "amdgpu\|^lib"; done
