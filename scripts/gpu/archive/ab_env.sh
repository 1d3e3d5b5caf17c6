#!/bin/bash
# A/B of an environment switch on one box: usage ab_env.sh VAR "v1 v2 v1 v2"
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in $2; do env $1=$v timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-families --no-roofline --no-variants 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1=$v', d['ms_per_step'])"; done
