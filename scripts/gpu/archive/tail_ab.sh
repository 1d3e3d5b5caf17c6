#!/bin/bash
# the fused tail of the residual blocks (LIDAL_TAIL_SUMS): tests, then A/B of the step
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/tail; mkdir -p $O
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_plan_gpu.py tests/test_teacher_forced_gpu.py tests/test_model_gpu.py -m gpu -q -x 2>&1 | tail -8
for rep in 1 2; do for t in 100000 0 1000000000; do for fr in 5 1; do
  LIDAL_TAIL_SUMS_ROWS=$t timeout 900 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $([ $fr = 5 ] && [ $rep = 1 ] || echo --no-families) --no-variants --no-secondary 2>$O/err.log > $O/line_${t}_${fr}_$rep.json; python - $O/line_${t}_${fr}_$rep.json $t $fr $rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f = d.get('families', {})
print('tail_sums %s frames %s rep %s: %.3f ms' % (sys.argv[2], sys.argv[3], sys.argv[4], d['ms_per_step']),
      {k: f[k]['ms'] for k in ('batch_norm', 'fused_elementwise', 'conv_apply') if k in f})
PY
done; done; done
