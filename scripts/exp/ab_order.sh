#!/bin/bash
# A/B of the level-0 voxel order (reference sorted-hash order vs lexicographic) on one box
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for o in hash lex hash lex; do LIDAL_L0_ORDER=$o timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-secondary --no-roofline --no-variants 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('order=$o', d['ms_per_step'], round(d['config']['loss'],4), {k:v.get('ms') for k,v in d.get('families',{}).items()})"; done
