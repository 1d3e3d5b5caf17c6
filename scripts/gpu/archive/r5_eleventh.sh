#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
for fr in "" 1; do
for lib in "" f32split0 f32split600; do
echo "== FRAME=$fr lib=$lib"
FRAME=$fr LIDAL_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/scripts/_abl/lib_$lib.so} timeout 600 python scripts/exp/split_check.py 2>&1 | grep -E "^s|dense"
done
done
