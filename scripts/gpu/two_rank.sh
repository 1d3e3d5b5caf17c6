#!/bin/bash
# the N>1 code path on ONE MI355X: two ranks on cuda:0 over gloo (bench.py --gpus 2)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/two_rank; mkdir -p $O
# GPU_MAX_HW_QUEUES=2: two processes with five streams each on ONE device oversubscribe its hardware queues -- measured
# (scripts/exp/ddp_probe.py, profiles/README.md round 4): backward passes of 1-13 s with the default 4 queues per process
# or with 8, 35 ms with 2.  One process per GPU (the real launch) is not affected.
GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline --no-families --no-variants > $O/bench2.log 2>&1; echo "rc=$?"; tail -1 $O/bench2.log | cut -c1-300
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -2
