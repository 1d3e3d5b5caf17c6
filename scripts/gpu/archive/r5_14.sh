#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_14; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
EXP_SHAPES=1:32:32,2:32:32,4:64:64,8:256:256,16:256:256 timeout 600 python scripts/exp_img.py 2>&1 | grep -E "^s"
EXP_SHAPES=1:32:32,2:32:32,4:64:64,8:256:256,16:256:256 LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_nomulti.so timeout 600 python scripts/exp_img.py 2>&1 | grep -E "^s" | sed 's/^/lean1 /'
done
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_model_gpu.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-variants > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_14/bench_line.json')))
print('ms/step', d['ms_per_step'])
print('families', {k: v.get('ms') for k, v in d['families'].items()})
PY
