#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_fourth; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/exp/tile_weights.py 2>&1 | tail -15
for rep in 1 2; do
EXP_SHAPES=1:96:96,1:32:32,4:128:128 timeout 600 python scripts/exp_img.py 2>&1 | grep -E "^s"
EXP_SHAPES=1:96:96,1:32:32,4:128:128 LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_rev.so timeout 600 python scripts/exp_img.py 2>&1 | grep -E "^s" | sed 's/^/rev /'
done
timeout 3000 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-families > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_fourth/bench_line.json')))
print('ms/step', d['ms_per_step'], 'value', d['value'])
print('variants', {k: (v.get('ms_per_step'), v.get('library_calls_per_step')) for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
print('roofline', d.get('roofline', {}).get('frac'), d.get('roofline', {}).get('launch_us'))
PY
