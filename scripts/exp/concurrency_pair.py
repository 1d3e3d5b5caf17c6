"""GPU experiment: the f64 fused block tail (lidal_add_relu_bwd_bn_sums + lidal_bn_bwd_from_sums) while f32 weight gradients
(lidal_conv_wgrad) run beside it on another stream -- the pair behind the run-to-run differences of the f32 mode
(profiles/README.md, round 5).  Each side's outputs are compared, bit for bit, with what it gives alone."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402

dev = torch.device('cuda')
L = B.lib()
ITERS = int(os.environ.get('ITERS', '300'))
batch = synth.make_train_batch(n_frames=2, n_points=60000, seed=100)
coords = torch.from_numpy(batch['coords_v_b']).to(dev)
with torch.enable_grad():
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
_ = kmap.koff
n = coords.shape[0]
g = torch.Generator(device='cpu').manual_seed(5)


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev)


# ---- the weight gradient (f32, 64 -> 64 on the level-0 map)
ci = co = 64
x, gy = rnd(n, ci), rnd(n, co)
slabs = int(L.lidal_conv_wgrad_slabs(n, n, 27, ci, co, 0))
partial = torch.empty(slabs * ci * co, dtype=torch.float32, device=dev)
gw = torch.empty(27, ci, co, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)


def wgrad(stream):
    B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0, B.ptr(gw),
                               B.ptr(partial), slabs, 27, ci, co, 0, stream), 'wgrad')


# ---- the tail (f32, n2 rows x c channels, with a shortcut BatchNorm)
n2, c = int(os.environ.get('ROWS', '50000')), 128
out, grad = rnd(n2, c), rnd(n2, c)
xa, xb = rnd(n2, c) * 1.5 + 0.3, rnd(n2, c) * 0.7 - 0.2
stats = []
for t in (xa, xb):
    stats.append((t.mean(0).contiguous(), (1.0 / torch.sqrt(t.var(0, unbiased=False) + 1e-5)).contiguous()))
wa, ba, wb, bb = rnd(c), rnd(c), rnd(c), rnd(c)
nb = L.lidal_bn_workspace_bytes(n2, c)
gm = torch.empty_like(out)
pa = torch.empty(nb, dtype=torch.uint8, device=dev)
pb = torch.empty(nb, dtype=torch.uint8, device=dev)
dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
gga, gba, ggb, gbb = (torch.empty(c, device=dev) for _ in range(4))


def tail(stream):
    B.check(L.lidal_add_relu_bwd_bn_sums(B.ptr(out), B.ptr(grad), B.ptr(gm), 0, n2, c, B.ptr(xa), B.ptr(stats[0][0]),
                                         B.ptr(stats[0][1]), B.ptr(pa), B.ptr(xb), B.ptr(stats[1][0]), B.ptr(stats[1][1]),
                                         B.ptr(pb), nb, stream), 'tail')
    B.check(L.lidal_bn_bwd_from_sums(B.ptr(xa), B.ptr(gm), c, 0, n2, c, B.ptr(wa), B.ptr(ba), 0, B.ptr(stats[0][0]),
                                     B.ptr(stats[0][1]), B.ptr(dxa), B.ptr(gga), B.ptr(gba), B.ptr(pa), nb, stream), 'bwd')
    B.check(L.lidal_bn_bwd_from_sums(B.ptr(xb), B.ptr(gm), c, 0, n2, c, B.ptr(wb), B.ptr(bb), 0, B.ptr(stats[1][0]),
                                     B.ptr(stats[1][1]), B.ptr(dxb), B.ptr(ggb), B.ptr(gbb), B.ptr(pb), nb, stream), 'bwd')


main = B.stream()
tail(main)
wgrad(main)
torch.cuda.synchronize()
ref_tail = [t.clone() for t in (gm, dxa, dxb, gga, gba, ggb, gbb)]
ref_gw = gw.clone()
names = ['gm', 'dx_a', 'dx_b', 'ggamma_a', 'gbeta_a', 'ggamma_b', 'gbeta_b']
for mode in ('alone', 'beside'):
    bad_tail = {k: 0 for k in names}
    bad_gw = 0
    for it in range(ITERS):
        if mode == 'beside':
            side.wait_stream(torch.cuda.current_stream())
            for _ in range(3):
                wgrad(side.cuda_stream)
            tail(main)
            torch.cuda.synchronize()
            bad_gw += int(not torch.equal(gw, ref_gw))
        else:
            tail(main)
            wgrad(main)
            torch.cuda.synchronize()
            bad_gw += int(not torch.equal(gw, ref_gw))
        for k, t, r in zip(names, (gm, dxa, dxb, gga, gba, ggb, gbb), ref_tail):
            if not torch.equal(t, r):
                bad_tail[k] += 1
                if bad_tail[k] == 1:
                    d = (t.double() - r.double()).abs()
                    print('   first difference in %s (%s, iteration %d): %d elements, max |d| %.3e at scale %.3e'
                          % (k, mode, it, int((d > 0).sum()), float(d.max()), float(r.abs().max())))
    print(mode, ': weight gradient differs in', bad_gw, 'of', ITERS, '; tail outputs:', bad_tail)
