#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c10; mkdir -p $O
timeout 400 python scripts/exp_img.py order > $O/exp_lean.log 2>&1; echo "lean rc=$?" >> $O/summary.txt
grep "^s\|^dense\|^global\|^block" $O/exp_lean.log
LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_w16.so timeout 400 python scripts/exp_img.py order > $O/exp_w16.log 2>&1; echo "w16 rc=$?" >> $O/summary.txt
grep "^s\|^dense\|^global\|^block" $O/exp_w16.log
