#!/bin/bash
# (a) is the stall of two gloo ranks on ONE device a matter of hardware queues?  (b) DDP over RCCL with one rank: the
# planned step + side streams beside RCCL's own stream in one process
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/exp/ddp_probe.py 2>&1 | grep "^step" | tail -5; }
run GPU_MAX_HW_QUEUES=8 FRAMES=1
run GPU_MAX_HW_QUEUES=2 FRAMES=1
run FRAMES=1
echo "== one rank, RCCL, DDP forced"
for ddp in 1 ""; do
BENCH_FORCE_DDP=$ddp MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force_ddp=$ddp', d['ms_per_step'], d['config'].get('parallelism'))"
BENCH_FORCE_DDP=$ddp MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --frames 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force_ddp=$ddp one scan', d['ms_per_step'])"
done
