"""The never-repeating input stream (bench.py: variants.fresh_stream) by itself: ms/step and the allocator's reserved
bytes / device allocations every 10 steps, under the allocator settings named in ALLOC (e.g.
ALLOC=roundup_power2_divisions:8 or ALLOC=expandable_segments:True; empty = torch's defaults)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

if os.environ.get('ALLOC'):
    torch.cuda.memory._set_allocator_settings(os.environ['ALLOC'])
if os.environ.get('THREADS'):
    torch.set_num_threads(int(os.environ['THREADS']))
import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
frames = int(os.environ.get('FRAMES', '5'))
steps = int(os.environ.get('STEPS', '60'))
res = bench.bench_fresh_stream(dev, 'spvcnn', 'bf16', frames, 120000, steps)
res.pop('what')
print(os.environ.get('ALLOC', 'default'), json.dumps(res))
