#!/bin/bash
# round 5: the multi-rank evidence with this round's library -- (a) `python bench.py --gpus 2` WITHOUT a launcher (the
# self-launching path), two gloo ranks on the one device; (b) one rank over RCCL with DataParallel and with torch DDP
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_multirank; mkdir -p $O
env -u WORLD_SIZE -u RANK -u LOCAL_RANK GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err; echo "2rank rc=$?"; cut -c1-400 $O/bench_2rank_gloo.json
BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --score-frames 16 --nei 10 > $O/bench_1rank_rccl_data_parallel.json 2> $O/dp.err; echo "dp rc=$?"; cut -c1-300 $O/bench_1rank_rccl_data_parallel.json
BENCH_TORCH_DDP=1 BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29545 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --no-secondary > $O/bench_1rank_rccl_torch_ddp.json 2> $O/ddp.err; echo "ddp rc=$?"; cut -c1-300 $O/bench_1rank_rccl_torch_ddp.json
