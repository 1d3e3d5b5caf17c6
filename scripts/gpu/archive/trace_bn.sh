#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r2c14; mkdir -p $O
cd /tmp
for v in "" bn512; do
  if [ -n "$v" ]; then export LIDAL_AMD_LIB=$GRAFT_REPO_ROOT/scripts/_abl/lib_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$v -- python3 $GRAFT_REPO_ROOT/scripts/exp_bn.py > $O/t_$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob('$O/t_$v/*/*kernel_trace.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'bn_' not in n: continue
    short=n.split('bn_')[1].split('_kernel')[0]
    acc[(short, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('LDS_Block_Size',''))].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print('variant [$v]')
for k,v in sorted(acc.items()):
    v=sorted(v); print('  %-14s grid %-8s  n=%3d  median %7.1f us' % (k[0],k[1],len(v),v[len(v)//2]/1e3))
PY
done
