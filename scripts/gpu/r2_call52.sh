#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for r in 0 50000 120000 250000 0 120000; do LIDAL_WGRAD_STREAM_ROWS=$r timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-families --no-roofline --no-variants 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('rows<=$r', d['ms_per_step'])"; done
