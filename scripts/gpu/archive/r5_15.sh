#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_15; mkdir -p $O
GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline > $O/torchrun2.json 2> $O/torchrun2.err; echo "torchrun rc=$?"; cut -c1-200 $O/torchrun2.json
env -u WORLD_SIZE -u RANK -u LOCAL_RANK GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline > $O/self2.json 2> $O/self2.err; echo "self rc=$?"; cut -c1-200 $O/self2.json
