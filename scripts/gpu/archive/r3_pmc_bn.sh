#!/bin/bash
# round 3: HBM traffic counters of the BatchNorm kernels on the level-0 shape (two --pmc passes, counters only)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BN_SHAPES=396662x96
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_bn; mkdir -p $O
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/scripts/exp_bn.py > $O/f.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/scripts/exp_bn.py > $O/w.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/pmc_bn bn_ > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-170
tail -3 $O/f.log
rm -rf $O/fetch $O/write
