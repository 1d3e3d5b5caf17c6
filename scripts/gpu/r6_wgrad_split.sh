#!/bin/bash
# round 6: the f32 weight gradient in the split form -- per layer against f64 and the exact kernel, the f32 tests, the f32 step
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_wgrad_split; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/exp/wgrad_split_check.py 2>&1 | grep -v amdgpu.ids | tee $O/layers.txt
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_teacher_forced_gpu.py tests/test_determinism_gpu.py tests/test_benchsize_gpu.py tests/test_scoring_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
timeout 900 python bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-variants --no-roofline > $O/f32.json 2> $O/f32.err; echo "f32 rc=$?"
python3 - <<PY
import json
d = json.load(open('$O/f32.json'))
print('f32 ms_per_step', d['ms_per_step'])
for k, v in d['families'].items():
    print('  %-20s %s' % (k, json.dumps(v)[:200]))
PY
