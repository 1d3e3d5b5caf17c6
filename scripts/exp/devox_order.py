"""Would devoxelize forward / voxelize backward gain from walking the points in CELL order (points of one coarse cell
next to each other: their 8 corner rows then come from L1 / L2 instead of 8 fresh rows per point)?  Same kernels, the
index / weight rows physically permuted into cell order (the output order changes with them: a timing experiment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, Geometry
from lidal_amd.nn import functional as F
from lidal_amd.nn.functional.invlist import inverse_lists
from exp_img import timeit

dev = 'cuda'
b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).train()
g = Geometry.build(model, coords, grad=True)
z = g.z
n = coords.shape[0]
for s, c in ((16, 256), (4, 128)):
    key = (s, s, s)
    pidx, counts = z.additional_features['idx_query'][key], z.additional_features['counts'][key]
    idx, w = z.idx_query[key], z.weights[key]
    m = counts.shape[0]
    order, _ = inverse_lists(pidx.int().contiguous(), m)
    order = order.long()
    idx_o, w_o, pidx_o = idx[order].contiguous(), w[order].contiguous(), pidx[order].contiguous()
    vox = torch.randn(m, c, device=dev).bfloat16()
    gv = torch.randn(m, c, device=dev).bfloat16()
    pf = torch.randn(n, c, device=dev).bfloat16().requires_grad_(True)
    t0 = timeit(lambda: F.spdevoxelize(vox, idx, w))
    t1 = timeit(lambda: F.spdevoxelize(vox, idx_o, w_o))
    y0 = F.spvoxelize(pf, pidx, counts)
    y1 = F.spvoxelize(pf, pidx_o, counts)
    b0 = timeit(lambda: y0.backward(gv, retain_graph=True))
    b1 = timeit(lambda: y1.backward(gv, retain_graph=True))
    print('stride %2d c %3d: devoxelize forward %6.1f us in point order, %6.1f in cell order;  voxelize backward %6.1f -> %6.1f'
          % (s, c, t0, t1, b0, b1))
