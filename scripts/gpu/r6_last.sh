#!/bin/bash
# round 6: the full -m gpu suite as the driver runs it (-x), then smoke(), then r6_final.sh
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/final6_$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu/r6_final.sh $TAG
