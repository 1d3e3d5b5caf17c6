#!/bin/bash
# round 6: the N>1 code paths with this round's library on ONE MI355X -- two gloo ranks on cuda:0 (bench.py --gpus 2, launched
# by a launcher and self-launched), one rank over RCCL, the 2-rank tests
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_multirank; mkdir -p $O
cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline --no-families --no-variants > $O/bench2.log 2>&1; echo "2-rank gloo rc=$?"; tail -1 $O/bench2.log > $O/bench_2rank_gloo_one_device.json; cut -c1-300 $O/bench_2rank_gloo_one_device.json
GPU_MAX_HW_QUEUES=2 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline --no-families --no-variants > $O/bench_2rank_gloo_one_device_self_launched.json 2> $O/self.err; echo "self-launched rc=$?"; cut -c1-300 $O/bench_2rank_gloo_one_device_self_launched.json
for fr in 5 1; do
BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --frames $fr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --score-frames 16 --nei 10 > $O/bench_1rank_rccl_${fr}scan.json 2> $O/err_$fr.log; echo "1-rank rccl $fr scans rc=$?"
cut -c1-300 $O/bench_1rank_rccl_${fr}scan.json
done
timeout 900 python -m pytest tests/test_multirank_gpu.py -m gpu -q 2>&1 | tail -2
