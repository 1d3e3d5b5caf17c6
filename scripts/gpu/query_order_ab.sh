#!/bin/bash
# queries of the inter-frame match walked in the query frame's cell order (LIDAL_QUERY_ORDER): suite, kernel time, frames/s
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/qorder; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for q in 1 0; do
  cd /tmp
  LIDAL_QUERY_ORDER=$q timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$q -- python3 $GRAFT_REPO_ROOT/scripts/profile_scoring.py 12 > $O/prof$q.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $O/p$q -name '*kernel_stats.csv' | head -1); echo "query order $q:"; grep "interframe" $f | cut -d, -f1-4 | cut -c1-160; rm -rf $O/p$q
done
for rep in 1 2; do for q in 1 0; do
  LIDAL_QUERY_ORDER=$q timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-families --no-variants 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep query order $q: %.3f ms' % d['ms_per_step'], {k: x['value'] for k, x in d['secondary']['by_nei'].items()})"
done; done
grep "bound to" $O/err.log
