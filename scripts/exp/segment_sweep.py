"""Ordered segment sums (voxelize forward, devoxelize backward) on the bench batch's real lists, per kernel form:
parts = 0 one wave per voxel, 1 / 2 / 4 workgroups per voxel.  Needs a hook `lidal_debug_set_segment_parts(int)` that
overrides voxel.hip's segment_parts() (round 3 had it for this measurement only; the table is in voxel.hip)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd import backend as B, synth
from lidal_amd.network import SPVCNN, Geometry
from lidal_amd.nn import functional as F
from exp_img import timeit

dev = 'cuda'
b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).train()
g = Geometry.build(model, coords, grad=True)
z = g.z
n = coords.shape[0]
setp = B.lib_handle().lidal_debug_set_segment_parts
setp.argtypes = [ctypes.c_int]
for s, c in ((16, 256), (4, 128), (1, 96), (1, 32)):
    key = (s, s, s)
    pidx, counts = z.additional_features['idx_query'][key], z.additional_features['counts'][key]
    idx, w = z.idx_query[key], z.weights[key]
    m = counts.shape[0]
    feats = torch.randn(n, c, device=dev).bfloat16()
    vox = torch.randn(m, c, device=dev).bfloat16().requires_grad_(True)
    gout = torch.randn(n, c, device=dev).bfloat16()
    out = F.spdevoxelize(vox, idx, w)
    line = 'stride %2d  c %3d  m %6d:' % (s, c, m)
    for parts in (-1, 0, 1, 2, 4):
        setp(parts)
        t_v = timeit(lambda: F.spvoxelize(feats, pidx, counts))
        t_d = timeit(lambda: out.backward(gout, retain_graph=True))
        line += '   parts %2d: vox fwd %6.1f  devox bwd %6.1f' % (parts, t_v, t_d)
    setp(-1)
    print(line)
