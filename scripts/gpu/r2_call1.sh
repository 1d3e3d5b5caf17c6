#!/bin/bash
# round 2, GPU call 1: tests (incl. the new 2-rank ones), 2-rank bench on one device, PMC diagnosis
# of the roofline kernel.  Everything lands in gpurun_out/r2c1/.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r2c1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
# 2 ranks on ONE device over gloo (torchrun forks before any GPU call)
BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --score-frames 6 \
  > $O/bench_2rank.log 2>&1; echo "2rank rc=$?" | tee -a $O/summary.txt
tail -2 $O/bench_2rank.log
rocprofv3 -L > $O/counters.txt 2>&1
cd /tmp
P="python3 $GRAFT_REPO_ROOT/bench.py --roofline-only"
R=$GRAFT_REPO_ROOT/$O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- $P > $R/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/pmc1 -- $P > $R/pmc1.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $R/pmc2 -- $P > $R/pmc2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $R/pmc3 -- $P > $R/pmc3.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $R/pmc4 -- $P > $R/pmc4.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/pmc5 -- $P > $R/pmc5.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $R/pmc6 -- $P > $R/pmc6.log 2>&1
cd $GRAFT_REPO_ROOT
# keep only the per-kernel rows of conv_apply from the (large) counter files
python3 scripts/gpu/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
ls -R $O | head -50
