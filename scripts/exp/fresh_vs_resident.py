"""Why is the changing-batches variant of bench.py slower than the resident batch?  Same bench_train on (a) the resident
batch, (b) a list of 8 copies of it, (c) 8 differently augmented batches; allocator statistics of the timed steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device('cuda')
batch = bench.make_batch(5, 120000, 7122, dev)
fresh = bench.make_fresh_batches(5, 120000, 7122, dev, 8)
copies = [tuple(t.clone() for t in batch) for _ in range(8)]
for name, b in (('resident batch', batch), ('8 copies of it', copies), ('8 different batches', fresh), ('resident batch', batch)):
    s0 = torch.cuda.memory_stats()
    r = bench.bench_train(1, 0, dev, 'spvcnn', 'bf16', b, 16, 8, ddp=False)
    s1 = torch.cuda.memory_stats()
    print('%-20s %.3f ms/step   voxels %d   device mallocs during the run %d   reserved %.1f GB' % (
        name, r['seconds'] / r['steps'] * 1e3, r['voxels'], s1['num_device_alloc'] - s0['num_device_alloc'],
        torch.cuda.memory_reserved() / 1e9), flush=True)
    del r
