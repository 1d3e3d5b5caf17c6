#!/bin/bash
# DDP over RCCL with ONE rank on one MI355X: the nccl branches of bench.py / sharding.py executed at all
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/rccl1; mkdir -p $O
for fr in 5 1; do
BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --frames $fr --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-families --no-variants --score-frames 16 --nei 10 > $O/line_$fr.json 2> $O/err_$fr.log; echo "rc=$?"
tail -5 $O/err_$fr.log; cut -c1-600 $O/line_$fr.json
done
