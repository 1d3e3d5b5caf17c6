"""The bf16 convolution of the layer shapes of the bench batch through lidal_conv_apply_image: time per launch and error
against an f64 reference on a sample of rows.  Run once per variant (LIDAL_LEAN32=0 / 1, or LIDAL_AMD_LIB=...): the
selection is read once per process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth
from lidal_amd.nn import functional as F
from lidal_amd.nn.functional.conv import _weight_image

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

dev = torch.device('cuda')
frames = int(os.environ.get('SCANS', '5'))
batch = synth.make_train_batch(n_frames=frames, n_points=120000, seed=7122)
coords = torch.from_numpy(batch['coords_v_b']).to(dev)
levels = {1: coords}
s = 1
while s < 16:
    levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
    s *= 2
L = B.lib()
print('variant: LIDAL_LEAN32=%s LIDAL_AMD_LIB=%s' % (os.environ.get('LIDAL_LEAN32'), os.environ.get('LIDAL_AMD_LIB')))
print('%-30s %10s %12s' % ('layer', 'us', 'err/scale'))
tot = 0.0
for stride, ci, co in [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 32, 64), (2, 64, 64), (4, 64, 128), (4, 128, 128), (4, 256, 128),
                       (8, 128, 256), (8, 256, 256), (8, 384, 256), (16, 256, 256)]:
    c = levels[stride]
    kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
    n = c.shape[0]
    o = kmap.order_out
    g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
    x = torch.randn(n, ci, generator=g).to(dev).bfloat16()
    w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
    with torch.no_grad():
        img = _weight_image(w, torch.bfloat16, n, 0)
    y = torch.empty((n, co), dtype=torch.bfloat16, device=dev)
    def launch():
        B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                         B.ptr(y), n, n, ci, co, 27, 0, B.BF16, None, None, 0, None, None, B.stream()), 'conv')
    t = timeit(launch)
    rows = min(20000, n)
    nbr = kmap.nbr_out[:, :rows].long()
    xd, wd = x.double(), w.bfloat16().double()
    ref = torch.zeros(rows, co, dtype=torch.float64, device=dev)
    for k in range(27):
        idx = nbr[k]
        m = idx >= 0
        ref[m] += xd[idx[m]] @ wd[k]
    err = (y[:rows].double() - ref).abs().max().item() / ref.abs().max().item()
    # the rows beyond the sample, and the tail of the last tile: a checksum against the reference of ALL rows in f32
    tot += t
    print('s%-2d %3d->%-3d (%7d rows)     %10.1f %12.2e' % (stride, ci, co, n, t, err), flush=True)
print('sum of the layers: %.1f us' % tot)
