"""lidal_ti_weights on the second stream with the inputs of a real geometry, beside (a) a real training step,
(b) a matmul, (c) nothing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, Geometry, glue
from lidal_amd.nn import functional as F
from lidal_amd.train_step import train_step

dev = torch.device('cuda')
b = synth.make_train_batch(n_frames=2, n_points=67000, seed=101)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
b2 = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
feats2, coords2, labels2 = (torch.from_numpy(b2[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b'))
torch.manual_seed(0)
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)
_real = F.ti_weights_and_index
cap = []
glue.F.ti_weights_and_index = lambda c, i, scale=1: (cap.append((c.clone(), i.clone(), scale)), _real(c, i, scale))[1]
Geometry.build(model, coords, grad=True)
glue.F.ti_weights_and_index = _real
torch.cuda.synchronize()
refs = [_real(c, i, s) for c, i, s in cap]
torch.cuda.synchronize()
g2 = Geometry.build(model, coords2, grad=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for busy in ('train_step', 'forward_only', 'matmul', 'none'):
    bad = 0
    cols = set()
    for it in range(60):
        if busy == 'train_step':
            train_step(model, opt, feats2, coords2, labels2, autocast=True, geometry=g2)
        elif busy == 'forward_only':
            from lidal_amd import SparseTensor
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                x = SparseTensor(feats2, coords2); x.geometry = g2
                model(x)
        elif busy == 'matmul':
            for _ in range(6):
                a @ a
        with torch.cuda.stream(side):
            outs = [_real(c, i, s) for c, i, s in cap]
        torch.cuda.synchronize()
        for (w, i32), (rw, ri) in zip(outs, refs):
            if not torch.equal(w, rw):
                bad += 1
                cols.update(torch.nonzero(w != rw)[:, 1].tolist())
    print('%-13s: %d of %d calls differ from the call alone; weight columns: %s' % (busy, bad, 60 * len(cap), sorted(cols)))
