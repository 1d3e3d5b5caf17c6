"""The f32 weight gradient in the split form (lidal_conv_wgrad with LIDAL_F32_SPLIT: wgrad_dma.hip wgrad_split_kernel) against
the exact f32 MFMA kernel and an f64 reference, on the layer shapes of the bench batch: error relative to the gradient's
scale, and time per call (split pass + products + slab reduction)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth
from lidal_amd.nn import functional as F

def timeit(fn, reps=10):
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
batch = synth.make_train_batch(n_frames=int(os.environ.get('SCANS', '5')), n_points=120000, seed=7122)
coords = torch.from_numpy(batch['coords_v_b']).to(dev)
levels = {1: coords}
s = 1
while s < 16:
    levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
    s *= 2
L = B.lib()
print('%-30s %10s %10s %12s %12s' % ('layer', 'exact us', 'split us', 'exact err', 'split err'))
tot = {'exact': 0.0, 'split': 0.0}
for stride, ci, co in [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 32, 64), (2, 64, 64), (4, 64, 128), (4, 128, 128), (4, 256, 128),
                       (8, 128, 256), (8, 256, 256), (8, 384, 256), (16, 256, 256)]:
    c = levels[stride]
    with torch.enable_grad():
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
    n = c.shape[0]
    g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
    x = torch.randn(n, ci, generator=g).to(dev)
    gy = (torch.randn(n, co, generator=g) * 0.1).to(dev)
    outs, times = {}, {}
    for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
        slabs = int(L.lidal_conv_wgrad_slabs(n, n, 27, ci, co, code))
        assert slabs > 0, (name, slabs)
        partial = torch.empty((slabs, ci, co), dtype=torch.float32, device=dev)
        gw = torch.empty((27, ci, co), dtype=torch.float32, device=dev)
        def launch():
            B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0, B.ptr(gw),
                                       B.ptr(partial), slabs, 27, ci, co, code, B.stream()), 'wgrad')
        times[name] = timeit(launch)
        tot[name] += times[name]
        outs[name] = gw.clone()
        launch()
        assert torch.equal(gw, outs[name]), 'not reproducible'
    nbmaps = kmap.nbmaps.long()
    koff = kmap.koff.cpu().tolist()
    ref = torch.zeros(27, ci, co, dtype=torch.float64, device=dev)
    xd, gd = x.double(), gy.double()
    for k in range(27):
        pr = nbmaps[koff[k]:koff[k + 1]]
        if pr.shape[0]:
            ref[k] = xd[pr[:, 0]].t() @ gd[pr[:, 1]]
    scale = ref.abs().max().item()
    errs = {kk: (v.double() - ref).abs().max().item() / scale for kk, v in outs.items()}
    print('s%-2d %3d->%-3d (%7d rows)     %10.1f %10.1f %12.2e %12.2e' % (stride, ci, co, n, times['exact'], times['split'], errs['exact'], errs['split']), flush=True)
print('sum: exact %.1f us, split %.1f us' % (tot['exact'], tot['split']))
# dense form (the point branch): [n, 128]^T [n, 96]
n = coords.shape[0]
g = torch.Generator(device='cpu').manual_seed(5)
x = torch.randn(n, 128, generator=g).to(dev)
gy = torch.randn(n, 96, generator=g).to(dev)
koff = torch.tensor([0, n], dtype=torch.int64, device=dev)
for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
    slabs = int(L.lidal_conv_wgrad_slabs(n, n, 1, 128, 96, code))
    partial = torch.empty((slabs, 128, 96), dtype=torch.float32, device=dev)
    gw = torch.empty((1, 128, 96), dtype=torch.float32, device=dev)
    def launch():
        B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, None, B.ptr(koff), 0, B.ptr(gw), B.ptr(partial), slabs, 1, 128, 96, code, B.stream()), 'wgrad')
    t = timeit(launch)
    ref = x.double().t() @ gy.double()
    print('dense 128x96 over %d rows %s: %.1f us, err %.2e' % (n, name, t, (gw[0].double() - ref).abs().max().item() / ref.abs().max().item()))
