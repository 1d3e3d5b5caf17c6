#!/bin/bash
# the data-parallel wrapper: 2-rank tests on one device, one rank over RCCL (wrapper vs torch DDP vs none), 2-rank gloo bench
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/dp; mkdir -p $O
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_geometry_gpu.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do for cfg in "none" "dp" "torch"; do for fr in 5 1; do
  case $cfg in none) E="";; dp) E="BENCH_FORCE_DDP=1";; torch) E="BENCH_FORCE_DDP=1 BENCH_TORCH_DDP=1";; esac
  env $E MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --frames $fr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2> $O/err_${cfg}_$fr.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep $cfg frames $fr: %.3f ms' % d['ms_per_step'])" || tail -5 $O/err_${cfg}_$fr.log
done; done; done
bash scripts/gpu/two_rank.sh
cp gpurun_out/two_rank/bench2.log $O/two_rank_bench2.log
