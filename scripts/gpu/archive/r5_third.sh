#!/bin/bash
# round 5, third GPU call: neighbour indices in LDS (product) against the round-4 form (lib_oldidx), stamps, full suite
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_third; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 600 python scripts/exp_img.py > $O/img_new_$rep.log 2>&1; echo "img new rc=$?"; grep -E "^s|dense" $O/img_new_$rep.log
LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_oldidx.so timeout 600 python scripts/exp_img.py > $O/img_old_$rep.log 2>&1; echo "img old rc=$?"; grep -E "^s|dense" $O/img_old_$rep.log
done
LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_stamps.so timeout 600 python scripts/exp/phase_stamps.py $O/phase_stamps_ldsidx.json > $O/stamps.log 2>&1; echo "stamps rc=$?"; grep -E "^s|issue|wait|reads|barrier|residency" $O/stamps.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_third/bench_line.json')))
print('ms/step', d['ms_per_step'], 'value', d['value'])
print('variants', {k: v.get('ms_per_step') for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
print('families', {k: v.get('ms') for k, v in d.get('families', {}).items() if isinstance(v, dict)})
print('roofline', d.get('roofline', {}).get('frac'), d.get('roofline', {}).get('launch_us'))
PY
