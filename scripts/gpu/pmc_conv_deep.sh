#!/bin/bash
# PMC wave-state / LDS counters of the lean conv kernels over scripts/exp_img.py's layer shapes
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_deep; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/scripts/exp_img.py > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/scripts/exp_img.py > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/pmc_deep conv_lean_kernelIDF16bLi8ELi128 > $O/summary.txt 2>&1; cat $O/summary.txt
python3 scripts/gpu/pmc_summary.py gpurun_out/pmc_deep conv_lean_kernelIDF16bLi6ELi192 >> $O/summary.txt 2>&1; tail -20 $O/summary.txt
find $O -name "*.csv" -size +2M -delete
