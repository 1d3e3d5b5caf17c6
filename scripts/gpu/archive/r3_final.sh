#!/bin/bash
# round 3: default bench line + the rocprof summaries that go to profiles/ (tag = $1)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/final3_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err; head -c 300 $O/bench_line.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants > $O/step.log 2>&1; echo "step rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/roof.log 2>&1; echo "roof rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $O/write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/final3_$TAG conv_ > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
for d in step roof score; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in fetch write; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 - "$f" "$O/${d}_counter_collection_conv.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_' in r.get('Kernel_Name', '')]
if rows:
    w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
done
rm -rf $O/step $O/roof $O/fetch $O/write $O/score
python3 scripts/gpu/stats_table.py $O/step_kernel_stats.csv 10 30
python3 scripts/gpu/stats_table.py $O/roof_kernel_stats.csv 1 5
ls -la $O
