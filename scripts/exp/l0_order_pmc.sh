#!/bin/bash
# PMC line traffic + durations of the level-0 96->96 conv / weight gradient under the two level-0 voxel orders
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/l0_pmc; mkdir -p $O
cd /tmp
for o in hash morton; do
  export LIDAL_L0_ORDER=$o
  python3 $GRAFT_REPO_ROOT/scripts/exp/l0_order_layer.py 2>&1 | tail -1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$o -- python3 $GRAFT_REPO_ROOT/scripts/exp/l0_order_layer.py > $O/t_$o.log 2>&1; echo "trace $o rc=$?"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$o -- python3 $GRAFT_REPO_ROOT/scripts/exp/l0_order_layer.py > $O/f_$o.log 2>&1; echo "fetch $o rc=$?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$o -- python3 $GRAFT_REPO_ROOT/scripts/exp/l0_order_layer.py > $O/w_$o.log 2>&1; echo "write $o rc=$?"
done
cd $GRAFT_REPO_ROOT
for o in hash morton; do
  echo "=== $o"
  for k in wgrad_dma_kernel conv_lean_kernel; do
    python3 scripts/gpu/pmc_summary.py gpurun_out/l0_pmc/fetch_$o $k | grep -v "^==" | head -3
    python3 scripts/gpu/pmc_summary.py gpurun_out/l0_pmc/write_$o $k | grep -v "^==" | head -3
  done
  f=$(find $O/trace_$o -name "*kernel_stats.csv" | head -1); grep -E "wgrad_dma_kernel|conv_lean_kernel|wgrad_dma_reduce" $f | cut -c1-60,100-260 | head -4
done > $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete
