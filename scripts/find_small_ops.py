"""Which Python lines launch the many tiny fill / copy / add kernels of a train step?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from lidal_amd import synth
from lidal_amd.network import SPVCNN
from lidal_amd.train_step import train_step
b = synth.make_train_batch(n_frames=2, n_points=120000, seed=7122)
dev = 'cuda'
coords = torch.from_numpy(b['coords_v_b']).to(dev); feats = torch.from_numpy(b['feats_v_b']).to(dev); labels = torch.from_numpy(b['labels_v_b']).to(dev)
model = SPVCNN(19).to(dev).train(); opt = torch.optim.Adam(model.parameters(), fused=True)
for _ in range(3): train_step(model, opt, feats, coords, labels, autocast=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
want = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::zeros', 'aten::_to_copy', 'aten::contiguous', 'aten::clone')
cnt = collections.Counter()
for e in prof.events():
    if e.name in want:
        st = [s for s in (e.stack or []) if 'lidal_amd' in s or 'train_step' in s or 'autograd' in s][:2]
        cnt[(e.name, ' <- '.join(x.strip()[-70:] for x in st))] += 1
for (k, v) in cnt.most_common(40):
    print(v, k[0], '|', k[1])
