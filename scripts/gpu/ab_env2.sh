#!/bin/bash
# A/B of an environment switch on one box: 5-scan step (prefetch and in-line) : usage ab_env2.sh VAR "v1 v2 v1 v2"
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in $2; do env $1=$v timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-roofline --no-families 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1=$v', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['variants'].items()})"; done
