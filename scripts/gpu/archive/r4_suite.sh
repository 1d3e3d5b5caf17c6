#!/bin/bash
# round 4: the whole GPU suite, then the N>1 code path on one device (two ranks over gloo)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/suite4; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 8 --nei 10 --no-cpu-baseline --no-families --no-variants > $O/bench2.log 2>&1; echo "two-rank rc=$?"; tail -1 $O/bench2.log | cut -c1-400
