#!/bin/bash
# which allocation site feeds the uninitialised read: scripts/exp/uninit_probe.py patched per source file
for only in "" plan.py backend.py geometry.py glue.py conv.py invlist.py voxelize.py devoxelize.py norm.py query.py hash.py tensor.py unet.py; do
  echo "== only [$only]"
  timeout 600 python3 scripts/exp/uninit_probe.py $only 2>&1 | grep "SPVCNN f32" | cut -c1-120
  LIDAL_PLAN=0 timeout 600 python3 scripts/exp/uninit_probe.py $only 2>&1 | grep "SPVCNN f32" | cut -c1-120
done
