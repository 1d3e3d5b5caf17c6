#!/bin/bash
# counter traffic of the whole 5-scan step (two --pmc passes, no tracing beside them)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r5_step_traffic; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/fetch -name '*counter_collection.csv' | head -1); w=$(find $O/write -name '*counter_collection.csv' | head -1)
python3 scripts/gpu/pmc_step_traffic.py $f $w > $O/step_traffic.txt 2>&1; cat $O/step_traffic.txt | cut -c1-170
rm -rf $O/fetch $O/write
