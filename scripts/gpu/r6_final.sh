#!/bin/bash
# round 6: the default bench line + every rocprof summary / counter record that goes to profiles/ (tag = $1)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/final6_$TAG; mkdir -p $O
B="python3 $GRAFT_REPO_ROOT/bench.py"
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
cd $GRAFT_REPO_ROOT
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err | cut -c1-200; head -c 300 $O/bench_line.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- $B --steps 7 --warmup 3 $QUIET > $O/step.log 2>&1; echo "step rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -- $B --roofline-only > $O/roof.log 2>&1; echo "roof rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --roofline-only > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --roofline-only > $O/write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score.log 2>&1; echo "score rc=$?"
SCORE_DTYPE=f32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/score32 -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/score32.log 2>&1; echo "score32 rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/sfetch -- $B --steps 8 --warmup 2 $QUIET > $O/sfetch.log 2>&1; echo "step fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/swrite -- $B --steps 8 --warmup 2 $QUIET > $O/swrite.log 2>&1; echo "step write rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/final6_$TAG conv_ > $O/pmc_summary.txt 2>&1; grep -c . $O/pmc_summary.txt
for d in step roof score score32; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in fetch write; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 - "$f" "$O/${d}_counter_collection_conv.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_' in r.get('Kernel_Name', '')]
if rows:
    w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
done
python3 scripts/gpu/pmc_record.py gpurun_out/final6_$TAG r06_$TAG "scripts/gpu/r6_final.sh $TAG" > $O/pmc_record.log 2>&1; tail -1 $O/pmc_record.log | cut -c1-300
cp profiles/r06_pmc_conv_apply.json $O/ 2>/dev/null
f=$(find $O/sfetch -name '*counter_collection.csv' | head -1); w=$(find $O/swrite -name '*counter_collection.csv' | head -1)
ROWS=$(python3 -c "import json; print(json.load(open('$O/bench_line.json'))['config']['voxels_per_step_per_gpu'])")
python3 scripts/gpu/pmc_step_traffic.py $f $w $ROWS bf16 $O/r06_pmc_step_traffic.json > $O/step_traffic.txt 2>&1; head -12 $O/step_traffic.txt | cut -c1-170
rm -rf $O/step $O/roof $O/fetch $O/write $O/score $O/score32 $O/sfetch $O/swrite
python3 scripts/gpu/stats_table.py $O/step_kernel_stats.csv 10 30
python3 scripts/gpu/stats_table.py $O/roof_kernel_stats.csv 1 5
for fr in 1 5; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t$fr -- $B --frames $fr --steps 12 --warmup 3 $QUIET > $O/steady$fr.log 2>&1; echo "steady $fr rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(find $O/t$fr -name "*kernel_trace.csv" | head -1)
  python3 scripts/gpu/steady_counts.py $f 60 > $O/steady_counts_${fr}scan.txt; head -4 $O/steady_counts_${fr}scan.txt
  rm -rf $O/t$fr
done
ls -la $O
