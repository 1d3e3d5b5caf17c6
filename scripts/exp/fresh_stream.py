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
st = torch.cuda.memory_stats(dev)
print({k: (st[k] >> 20) for k in ('reserved_bytes.large_pool.current', 'reserved_bytes.small_pool.current',
                                   'active_bytes.all.peak', 'inactive_split_bytes.all.current',
                                   'reserved_bytes.all.peak')}, 'MB;',
      {k: st[k] for k in ('segment.large_pool.current', 'segment.small_pool.current', 'num_alloc_retries')})

print({k: (st[k] >> 20) for k in ('active_bytes.all.current', 'allocated_bytes.all.current', 'reserved_bytes.all.current')}, 'MB current')
import collections
free = collections.Counter()
used = collections.Counter()
for seg in torch.cuda.memory_snapshot():
    for blk in seg['blocks']:
        (free if blk['state'] == 'inactive' else used)[(seg.get('stream', 0), blk['size'] >> 20)] += 1
tot_free = sum(k[1] * v for k, v in free.items())
print('cached free blocks: %d MB in %d blocks; by (stream, MB) x count, largest first:' % (tot_free, sum(free.values())))
print(sorted(((k[1] * v, k, v) for k, v in free.items()), reverse=True)[:25])
print('live blocks by (stream, MB):', sorted(((k[1] * v, k, v) for k, v in used.items()), reverse=True)[:15])
