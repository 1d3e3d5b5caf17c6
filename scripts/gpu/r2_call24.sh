#!/bin/bash
# PMC wave-state counters of the DMA weight-gradient kernel (96->96 and 32->32, level 0)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 LIDAL_EXP_ORDERS=dataset
O=$GRAFT_REPO_ROOT/gpurun_out/r2c24; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/scripts/exp_memorder.py > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/scripts/exp_memorder.py > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc3 -- python3 $GRAFT_REPO_ROOT/scripts/exp_memorder.py > $O/p3.log 2>&1; echo "p3 rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/r2c24 wgrad_dma > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete
