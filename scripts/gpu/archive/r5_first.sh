#!/bin/bash
# round 5, first GPU call: the new tests, then the default bench line (f32 secondary, CPU baselines per protocol)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_first; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x -k "one_weight_only or two_models_training or starts_its_own_ranks or bf16_inference_against_f32 or two_rank_ddp or merges_inside or one_json_line" -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5_first/bench_line.json')))
print('ms/step', d['ms_per_step'], 'value', d['value'])
print('secondary', json.dumps(d.get('secondary'))[:1500])
print('cpu', d.get('cpu_baseline'))
print('variants', {k: v.get('ms_per_step') for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
print('families', {k: v.get('ms') for k, v in d.get('families', {}).items() if isinstance(v, dict)})
print('roofline', d.get('roofline', {}).get('frac'), d.get('roofline', {}).get('launch_us'))
PY
