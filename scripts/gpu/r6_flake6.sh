#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_flake6; mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "A=1" "LIDAL_PLAN_SIDE_F32=0" "LIDAL_PLAN_F32_BN_ALONE=0"; do
  env $cfg timeout 600 python bench.py --dtype f32 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-variants --no-roofline --no-families > $O/t.json 2> $O/t.err
  echo "$cfg: f32 ms_per_step $(python3 -c "import json; print(json.load(open('$O/t.json'))['ms_per_step'])")"; tail -1 $O/t.err | cut -c1-200
done
timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_plan_gpu.py -q -m gpu 2>&1 | tail -4
