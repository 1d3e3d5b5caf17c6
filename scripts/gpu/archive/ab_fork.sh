#!/bin/bash
# A/B of the LIDAL_FORK bits on one box: usage ab_fork.sh "7 15 7 15"
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for f in ${1:-7 15 7 15}; do LIDAL_FORK=$f timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-families --no-roofline --no-variants 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fork=$f', d['ms_per_step'])"; done
