#!/bin/bash
# round 3: wave-state / instruction-mix counters of the convolution kernels over scripts/exp_img.py's layer shapes
# (two --pmc passes, counters only: no trace domains)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/pmc3; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc1 -- python3 $GRAFT_REPO_ROOT/scripts/exp_img.py > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2 -- python3 $GRAFT_REPO_ROOT/scripts/exp_img.py > $O/p2.log 2>&1; echo "p2 rc=$?"
cd $GRAFT_REPO_ROOT
for k in conv_lean_kernelIDF16bLi6ELi192 conv_lean_deep_kernelIDF16bLi8ELi128 conv_lean_kernelIDF16bLi4ELi128 conv_lean_kernelIDF16bLi8ELi128 conv_lean_kernelIDF16bLi2ELi64; do
  python3 scripts/gpu/pmc_summary.py gpurun_out/pmc3 $k
done > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-150
tail -12 $O/p1.log
rm -rf $O/pmc1 $O/pmc2
