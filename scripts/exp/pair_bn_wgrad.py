"""GPU experiment (round 6): the f32 BatchNorm backward (lidal_bn_bwd: f64 sums) while a weight gradient runs beside it on a
second stream -- exact f32 MFMA, the split form (bf16 MFMA + LDS-DMA + transposed LDS reads), or the bf16 kernel.  Inside
the planned f32 step the pair (BatchNorm backward, split-form weight gradient) gives gradients that differ run to run;
does the pair do it in isolation?  Every output of either side is compared, bit for bit, with what it gives alone."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402

dev = torch.device('cuda')
L = B.lib()
ITERS = int(os.environ.get('ITERS', '200'))
batch = synth.make_train_batch(n_frames=2, n_points=60000, seed=100)
coords = torch.from_numpy(batch['coords_v_b']).to(dev)
with torch.enable_grad():
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
_ = kmap.koff
n = coords.shape[0]
g = torch.Generator(device='cpu').manual_seed(5)


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev)


ci = co = 96
x, gy = rnd(n, ci), rnd(n, co) * 0.1
xb16, gyb16 = x.bfloat16(), gy.bfloat16()
side = torch.cuda.Stream(device=dev)


def make_wgrad(code, a, b):
    slabs = int(L.lidal_conv_wgrad_slabs(n, n, 27, ci, co, code))
    partial = torch.empty(slabs * ci * co, dtype=torch.float32, device=dev)
    gw = torch.empty(27, ci, co, dtype=torch.float32, device=dev)

    def run(stream):
        B.check(L.lidal_conv_wgrad(B.ptr(a), B.ptr(b), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0, B.ptr(gw),
                                   B.ptr(partial), slabs, 27, ci, co, code, stream), 'wgrad')
    return run, gw


AGGR = {'exact f32 wgrad': make_wgrad(B.F32, x, gy), 'split-form wgrad': make_wgrad(B.F32_SPLIT, x, gy),
        'bf16 wgrad': make_wgrad(B.BF16, xb16, gyb16)}

# ---- the victim: BatchNorm backward over n rows x c channels, f32 (f64 partial sums)
c = 96
xa, dy = rnd(n, c) * 1.5 + 0.3, rnd(n, c) * 0.01
mean, invstd = xa.mean(0).contiguous(), (1.0 / torch.sqrt(xa.var(0, unbiased=False) + 1e-5)).contiguous()
wa, ba = rnd(c), rnd(c)
nb = L.lidal_bn_workspace_bytes(n, c)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
dx = torch.empty_like(xa)
gga, gba = torch.empty(c, device=dev), torch.empty(c, device=dev)


def bn(stream, relu=1):
    B.check(L.lidal_bn_bwd(B.ptr(xa), B.ptr(dy), c, 0, n, c, B.ptr(wa), B.ptr(ba), relu, B.ptr(mean), B.ptr(invstd), B.ptr(dx),
                           B.ptr(gga), B.ptr(gba), B.ptr(ws), nb, stream), 'bn_bwd')


main = B.stream()
bn(main)
torch.cuda.synchronize()
ref_bn = [t.clone() for t in (dx, gga, gba)]
names = ['dx', 'ggamma', 'gbeta']
for aname, (wg, gw) in AGGR.items():
    wg(main)
    torch.cuda.synchronize()
    ref_gw = gw.clone()
    for mode in ('alone', 'beside'):
        bad = {k: 0 for k in names}
        bad_gw = 0
        worst = 0.0
        for it in range(ITERS):
            dx.fill_(float('nan')); gga.fill_(float('nan')); gba.fill_(float('nan'))
            gw.fill_(float('nan'))
            torch.cuda.synchronize()
            if mode == 'beside':
                side.wait_stream(torch.cuda.current_stream())
                wg(side.cuda_stream)
                bn(main)
            else:
                bn(main)
                torch.cuda.synchronize()
                wg(main)
            torch.cuda.synchronize()
            for k, t, r in zip(names, (dx, gga, gba), ref_bn):
                if not torch.equal(t, r):
                    bad[k] += 1
                    worst = max(worst, float((t.double() - r.double()).abs().max() / r.double().abs().max()))
            if not torch.equal(gw, ref_gw):
                bad_gw += 1
        print('%-18s %-7s: of %d iterations the BatchNorm backward differed in %s (largest relative difference %.2e); the weight '
              'gradient differed in %d' % (aname, mode, ITERS, bad, worst, bad_gw), flush=True)
