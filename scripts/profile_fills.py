"""GPU: who launches the small fill / copy / memset operators of a train step?  Walks the CPU-side
parent chain of every aten::fill_ / zero_ / copy_ / _to_copy event of one profiled step."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402


def main():
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        train_step(model, opt, feats, coords, labels, autocast=True)
        torch.cuda.synchronize()
    want = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::zeros_like', 'aten::full',
            'aten::ones', 'aten::clone', 'aten::contiguous', 'aten::add', 'aten::add_', 'aten::cat')
    chains = Counter()
    for e in prof.events():
        if e.name in want and getattr(e, 'device_type', None) is not None:
            chain, p = [], e.cpu_parent
            while p is not None:
                chain.append(p.name)
                p = p.cpu_parent
            shapes = str(getattr(e, 'input_shapes', ''))[:60]
            chains[(e.name, ' <- '.join(chain[:4]), shapes)] += 1
    for (name, chain, shapes), c in sorted(chains.items(), key=lambda kv: -kv[1])[:60]:
        print('%4d  %-16s %-60s %s' % (c, name, shapes, chain))


if __name__ == '__main__':
    main()
