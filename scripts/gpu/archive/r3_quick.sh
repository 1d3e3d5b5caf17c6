#!/bin/bash
# usage: r3_quick.sh TAG 'pytest -k expression' [extra python scripts...]
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r3_${1:-x}; mkdir -p $O
if [ -n "${2:-}" ]; then timeout 1500 python -m pytest tests -m gpu -q -k "$2" > $O/tests.log 2>&1; tail -25 $O/tests.log; fi
shift; shift
for sc in "$@"; do echo "== $sc"; timeout 900 python $sc 2>&1 | grep -v amdgpu.ids | tail -60; done
