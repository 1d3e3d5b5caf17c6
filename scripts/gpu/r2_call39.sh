#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in dmap_a8 dmap_a4 dmap_a2 dmap_a12; do echo "== $v"; LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so timeout 300 python scripts/exp_img.py 2>&1 | grep -v amdgpu | grep "s1   96\|s1   32\|s8  256"; done
