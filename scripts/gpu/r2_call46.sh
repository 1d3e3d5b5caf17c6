#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "" mw5 mw6; do echo "== ${v:-shipped}"; if [ -n "$v" ]; then export LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so; fi; timeout 300 python scripts/exp_img.py 2>&1 | grep -v amdgpu | grep "^s\|^dense"; done
