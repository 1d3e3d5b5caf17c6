"""Which kernel family, running on the main stream, disturbs lidal_ti_weights on the second stream?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lidal_amd
from lidal_amd import SparseTensor, synth
from lidal_amd import nn as spnn
from lidal_amd.network import SPVCNN, Geometry, glue
from lidal_amd.nn import functional as F

dev = torch.device('cuda')
b = synth.make_train_batch(n_frames=2, n_points=67000, seed=101)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
b2 = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords2 = torch.from_numpy(b2['coords_v_b']).to(dev)
torch.manual_seed(0)
model = SPVCNN(19).to(dev).train()
_real = F.ti_weights_and_index
cap = []
glue.F.ti_weights_and_index = lambda c, i, scale=1: (cap.append((c.clone(), i.clone(), scale)), _real(c, i, scale))[1]
Geometry.build(model, coords, grad=True)
glue.F.ti_weights_and_index = _real
torch.cuda.synchronize()
refs = [_real(c, i, s) for c, i, s in cap]
g2 = Geometry.build(model, coords2, grad=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()


def level(stride, c):
    st = (stride,) * 3
    cs = g2.x0.cmaps[st]
    x = SparseTensor(torch.randn(cs.shape[0], c, device=dev).bfloat16(), cs, stride)
    x.cmaps, x.kmaps = g2.x0.cmaps, g2.x0.kmaps
    return x


def conv_job(stride, ci, co, k=3, grad=False):
    conv = spnn.Conv3d(ci, co, k).to(dev)
    x = level(stride, ci)
    if grad:
        x.F.requires_grad_(True)

    def run():
        with torch.autocast('cuda', dtype=torch.bfloat16), torch.set_grad_enabled(grad):
            y = conv(x)
            if grad:
                y.F.float().sum().backward()
    return run


def bn_job(stride, c):
    bn = spnn.BatchNorm(c).to(dev).train()
    x = level(stride, c)

    def run():
        with torch.no_grad():
            bn(x)
    return run


def devox_job(stride, c):
    z = g2.z
    idx, w = z.idx_query[(stride,) * 3], z.weights[(stride,) * 3]
    f = torch.randn(g2.x0.cmaps[(stride,) * 3].shape[0], c, device=dev).bfloat16()
    return lambda: F.spdevoxelize(f, idx, w)


jobs = {
    'conv 96->96 k3 s1': conv_job(1, 96, 96),
    'conv 32->32 k3 s1': conv_job(1, 32, 32),
    'conv 256->256 k3 s8 (deep)': conv_job(8, 256, 256),
    'conv 256->256 k3 s16': conv_job(16, 256, 256),
    'conv 64->64 k3 s4': conv_job(4, 64, 64),
    'dense 96->96': conv_job(1, 96, 96, k=1),
    'conv 96->96 k3 s1 fwd+bwd': conv_job(1, 96, 96, grad=True),
    'batchnorm 96 s1': bn_job(1, 96),
    'devoxelize 256 s16': devox_job(16, 256),
}
for name, job in jobs.items():
    for _ in range(3):
        job()
    torch.cuda.synchronize()
    bad = 0
    for it in range(40):
        for _ in range(6):
            job()
        with torch.cuda.stream(side):
            outs = [_real(c, i, s) for c, i, s in cap]
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(w, rw)) for (w, _), (rw, _) in zip(outs, refs))
    print('%-30s: %3d of %d calls differ' % (name, bad, 40 * len(cap)), flush=True)
