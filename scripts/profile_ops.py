"""GPU: torch.profiler view of one train step -- which ATen operators (and which of their kernels)
the step spends GPU time in OUTSIDE liblidal_amd's own launches: fills, copies, adds, cats, the
optimizer.  usage: profile_ops.py [steps]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda')
    b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(b['coords_v_b']).to(dev)
    feats = torch.from_numpy(b['feats_v_b']).to(dev)
    labels = torch.from_numpy(b['labels_v_b']).to(dev)
    torch.manual_seed(7122)
    model = SPVCNN(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    for _ in range(3):
        train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(os.environ.get('PROFILE_STACKS'))) as prof:
        for _ in range(steps):
            train_step(model, opt, feats, coords, labels, autocast=True)
        torch.cuda.synchronize()
    ka = prof.key_averages()
    rows = [(e.key, e.count, getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)))
            for e in ka]
    rows = [r for r in rows if r[2] > 0]
    rows.sort(key=lambda r: -r[2])
    print('%-90s %8s %12s' % ('operator / kernel (self GPU time)', 'calls/st', 'ms/step'))
    for k, c, t in rows[:70]:
        print('%-90s %8.1f %12.3f' % (k[:90], c / steps, t / 1e3 / steps))


    if os.environ.get('PROFILE_STACKS'):        # who calls the small ATen operators
        want = ('aten::zero_', 'aten::fill_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::zeros',
                'aten::to', 'aten::contiguous', 'aten::clone')
        for e in prof.key_averages(group_by_stack_n=8):
            if e.key in want and e.count >= steps:
                frames = [f for f in e.stack if 'lidal_amd' in f or 'train_step' in f or 'bench' in f][:3]
                print('%-14s %6.1f/step  %s' % (e.key, e.count / steps, ' <- '.join(f.strip()[-70:] for f in frames)))


if __name__ == '__main__':
    main()
