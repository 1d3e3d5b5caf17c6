#!/bin/bash
# rocprofv3 summaries for profiles/: kernel stats of the train step, of the roofline-only run, and
# the FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic.  usage: r2_prof.sh TAG
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-x}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/step.log 2>&1; echo "step rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/roof.log 2>&1; echo "roof rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/prof_$TAG conv_ > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.csv" -size +3M -delete
tail -2 $O/roof.log
