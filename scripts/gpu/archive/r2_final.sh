#!/bin/bash
# full default bench + rocprof summaries for profiles/ (tag = $1)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-c}
O=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err; head -c 400 $O/bench_line.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/step.log 2>&1; echo "step rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/roof.log 2>&1; echo "roof rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/final_$TAG conv_ > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.csv" -size +3M -delete
