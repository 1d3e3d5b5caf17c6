#!/bin/bash
# round 6: wave-state counters of the two forms of the f32 split convolution on an 8-view frame (scripts/exp/split_check.py)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_split_pmc; mkdir -p $O
for form in 16 32; do
  cd /tmp
  export LIDAL_SPLIT_MFMA=$form FRAME=1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/f$form/pmc1 -- python3 $GRAFT_REPO_ROOT/scripts/exp/split_check.py > $O/p1_$form.log 2>&1; echo "p1 rc=$?"
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/f$form/pmc2 -- python3 $GRAFT_REPO_ROOT/scripts/exp/split_check.py > $O/p2_$form.log 2>&1; echo "p2 rc=$?"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f$form/stats -- python3 $GRAFT_REPO_ROOT/scripts/exp/split_check.py > $O/p3_$form.log 2>&1; echo "p3 rc=$?"
  cd $GRAFT_REPO_ROOT
  python3 scripts/gpu/pmc_summary.py gpurun_out/r6_split_pmc/f$form conv_split > $O/summary_$form.txt 2>&1
  f=$(find $O/f$form/stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep conv_split "$f" > $O/stats_$form.csv
  find $O/f$form -name "*.csv" -size +1M -delete
done
unset LIDAL_SPLIT_MFMA FRAME
cat $O/summary_16.txt $O/summary_32.txt | grep -v "^==" | sort | head -150
cat $O/stats_16.csv $O/stats_32.csv
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_scoring_gpu.py -x -q -m gpu -k "256" 2>&1 | tail -5
timeout 1500 python bench.py --steps 10 --warmup 3 --no-families > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); s=d['secondary']
print(d['ms_per_step'], s.get('value'), json.dumps(s.get('roofline')), json.dumps(s['by_dtype'].get('f32_exact')), json.dumps(s['by_dtype']['bf16'].get('roofline')))
print(json.dumps(d['variants'].get('score_256')))
print(json.dumps(d['roofline']))
"
