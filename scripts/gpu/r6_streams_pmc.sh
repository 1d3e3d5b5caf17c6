#!/bin/bash
# round 6: fabric traffic and wave-state counters of the streamed weight gradient beside the offset-major one
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams_pmc; mkdir -p $O
export LEVEL=${LEVEL:-0} BLOCK=${BLOCK:-1024} W=512 REPS=5 LIDAL_X_STREAM_DEPTH=${LIDAL_X_STREAM_DEPTH:-1}
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/scripts/exp/wgrad_streams.py > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/scripts/exp/wgrad_streams.py > $O/p2.log 2>&1; echo "p2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $GRAFT_REPO_ROOT/scripts/exp/wgrad_streams.py > $O/p3.log 2>&1; echo "p3 rc=$?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/tcc -- python3 $GRAFT_REPO_ROOT/scripts/exp/wgrad_streams.py > $O/p4.log 2>&1; echo "p4 rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/gpu/pmc_summary.py gpurun_out/r6_streams_pmc wgrad_ > $O/summary.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $O/summary.txt | cut -c1-200
tail -3 $O/p4.log
