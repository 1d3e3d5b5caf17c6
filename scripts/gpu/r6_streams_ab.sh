#!/bin/bash
# round 6: the 5-scan step with the weight gradients on rule streams (default: levels >= 150 k rows) against without
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams_ab; mkdir -p $O
cd $GRAFT_REPO_ROOT
QUIET="--no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants"
for rep in 1 2; do
  for rows in 0 150000 100000; do
    LIDAL_WGRAD_STREAMS_ROWS=$rows timeout 600 python bench.py --steps 30 --warmup 8 $QUIET > $O/b_${rows}_$rep.json 2> $O/b_${rows}_$rep.err
    python3 -c "
import json; d=json.load(open('$O/b_${rows}_$rep.json')); print('rows>=$rows rep $rep: ms_per_step', d['ms_per_step'], 'value', d['value'])"
  done
done
LIDAL_WGRAD_STREAMS_ROWS=0 timeout 900 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline --no-variants > $O/fam_off.json 2>$O/fam_off.err
timeout 900 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary --no-roofline --no-variants > $O/fam_on.json 2>$O/fam_on.err
python3 -c "
import json
for t in ('off','on'):
    d=json.load(open('$O/fam_%s.json'%t)); f=d['families']
    print(t, d['ms_per_step'], {k:(v.get('ms'), v.get('hbm_frac')) for k,v in f.items() if isinstance(v,dict) and 'ms' in v})
"
