#!/bin/bash
# the data-parallel wrapper on one rank over RCCL: wrapper vs torch DDP vs none; 2-rank tests
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/dp; mkdir -p $O
timeout 1200 python -m pytest tests/test_multirank_gpu.py -m gpu -q -x 2>&1 | tail -3
port=29550
for rep in 1 2; do for cfg in "none" "dp"; do for fr in 5 1; do
  case $cfg in none) E="";; dp) E="BENCH_FORCE_DDP=1";; torch) E="BENCH_FORCE_DDP=1 BENCH_TORCH_DDP=1";; esac
  port=$((port+1))
  env $E MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --frames $fr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2> $O/err_${cfg}_$fr.log > $O/line_${cfg}_${fr}_$rep.json; echo "rep $rep $cfg frames $fr rc=$? $(python -c "
import json,sys
d=json.loads(open('$O/line_${cfg}_${fr}_$rep.json').read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")"
done; done; done
