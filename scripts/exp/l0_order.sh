#!/bin/bash
# level-0 voxel order experiment: families of the 5-scan step under LIDAL_L0_ORDER=hash|morton
# (bench.py does not import the hook: run it as  python -c "import sys; sys.path.insert(0, 'scripts/exp'); import l0_morton; import runpy; ..."
#  -- the numbers in profiles/README.md were taken when the switch still lived in network/glue.py, commit d2e1be6..9008c12)
for o in hash morton; do
  LIDAL_L0_ORDER=$o python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-variants --no-roofline > gpurun_out/l0_$o.json 2> gpurun_out/l0_$o.err
  python - <<PY
import json
d = json.load(open("gpurun_out/l0_$o.json"))
f = d.get("families", {})
print("$o", d["ms_per_step"], {k: f[k]["ms"] for k in ("conv_apply", "conv_wgrad", "batch_norm", "point_voxel", "kernel_maps") if k in f}, f.get("whole_step", {}).get("ms"))
PY
done
