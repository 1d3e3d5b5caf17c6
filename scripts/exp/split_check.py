"""The split form of the f32 convolution (LIDAL_F32_SPLIT: conv_split_kernel) against the exact-f32 kernel and an f64
reference, on the layer shapes of the bench batch: error relative to the output scale, and time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth
from lidal_amd.nn import functional as F
from lidal_amd.nn.functional.conv import _weight_image

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
if os.environ.get('FRAME'):       # an 8-view inference frame instead of the 5-scan train batch
    import numpy as np
    seq = synth.make_sequence(1, n_points=120000, seed=7122)[0]
    coords = torch.from_numpy(synth.make_score_batch(seq['points'], seq['intensity'], np.random.default_rng(1), inf_reps=8)['coords_v_b']).to(dev)
else:
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
levels = {1: coords}
s = 1
while s < 16:
    levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
    s *= 2
L = B.lib()
print('%-28s %10s %10s %12s %12s' % ('layer', 'exact us', 'split us', 'exact err', 'split err'))
for stride, ci, co in [(1, 96, 96), (2, 32, 64), (4, 64, 128), (4, 128, 128), (8, 128, 256), (8, 256, 256), (8, 384, 256), (16, 256, 256)]:
    c = levels[stride]
    kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
    n = c.shape[0]
    o = kmap.order_out
    g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
    x = torch.randn(n, ci, generator=g).to(dev)
    w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
    outs = {}
    times = {}
    for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
        with torch.no_grad():
            img = _weight_image(w, torch.float32, n, 0, code)
        y = torch.empty((n, co), dtype=torch.float32, device=dev)
        wsb = int(L.lidal_conv_apply_workspace_bytes(n, co))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        def launch():
            B.check(L.lidal_conv_apply_image_ws(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                                B.ptr(y), n, n, ci, co, 27, 0, code, None, None, 0, None, None,
                                                B.ptr(ws) if wsb else None, wsb, B.stream()), 'conv')
        times[name] = timeit(launch)
        outs[name] = y.clone()
    # f64 reference on a sample of output rows
    rows = min(20000, n)
    nbr = kmap.nbr_out[:, :rows].long()
    xd, wd = x.double(), w.double()
    ref = torch.zeros(rows, co, dtype=torch.float64, device=dev)
    for k in range(27):
        idx = nbr[k]
        m = idx >= 0
        ref[m] += xd[idx[m]] @ wd[k]
    scale = ref.abs().max().item()
    errs = {k: ((v[:rows].double() - ref).abs().max().item() / scale) for k, v in outs.items()}
    print('s%-2d %3d->%-3d (%6d rows)     %10.1f %10.1f %12.2e %12.2e' % (stride, ci, co, n, times['exact'], times['split'], errs['exact'], errs['split']), flush=True)
# dense form
n = coords.shape[0]
g = torch.Generator(device='cpu').manual_seed(5)
x = torch.randn(n, 128, generator=g).to(dev)
w = (torch.randn(1, 128, 96, generator=g) * 0.05).to(dev)
for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
    with torch.no_grad():
        img = _weight_image(w, torch.float32, n, 0, code)
    y = torch.empty((n, 96), dtype=torch.float32, device=dev)
    def launch():
        B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), None, None, None, B.ptr(y), n, n, 128, 96, 1, 0, code, None, None, 0, None, None, B.stream()), 'conv')
    t = timeit(launch)
    ref = x[:20000].double() @ w[0].double()
    print('dense 128->96 %s: %.1f us, err %.2e' % (name, t, (y[:20000].double() - ref).abs().max().item() / ref.abs().max().item()))
