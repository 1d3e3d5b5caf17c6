"""The drop-in surface step (bench.py variants.dropin_surface) alone, for rocprofv3 --kernel-trace --stats:
    python scripts/exp/dropin_profile.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
batch = bench.make_batch(5, 120000, 7122, dev)
r = bench.bench_dropin_surface(dev, os.environ.get('MODEL', 'spvcnn'), os.environ.get('DTYPE', 'bf16'), batch, steps, 2)
print(r)
