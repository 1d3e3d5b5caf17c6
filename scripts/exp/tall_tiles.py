"""GPU experiment: the wide / coarse convolution layers under taller per-wave tiles (lidal_debug_set)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from exp_img import make_image, timeit  # noqa: E402
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402


def main():
    dev = torch.device('cuda')
    lib = B.lib_handle()
    dbg = lib.lidal_debug_set
    dbg.argtypes = [ctypes.c_int, ctypes.c_int]
    batch = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '5')), n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    shapes = [(4, 128, 128), (4, 192, 128), (8, 256, 256), (8, 384, 256), (8, 128, 128), (16, 256, 256), (16, 256, 128)]
    print('%-26s' % 'layer', ''.join('%16s' % c for c in ('lean nb*', 'lean nb4', 'lean nb8', 't1 4x2 nb4', 't2 2x4 nb4',
                                                        't2 2x4 nb8', 't3 2x2 nb4', 't4 4x1 nb4', 't3 2x2 nb8')))
    for stride, ci, co in shapes:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
        x = torch.randn(n, ci, generator=g).to(dev).bfloat16()
        w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
        o = kmap.order_out
        out = torch.empty((n, co), dtype=torch.bfloat16, device=dev)
        res, ref = [], None
        for nb, tall in ((0, 0), (4, 0), (8, 0), (4, 1), (4, 2), (8, 2), (4, 3), (4, 4), (8, 3)):
            dbg(0, nb)
            dbg(1, tall)
            img = make_image(w, torch.bfloat16, n)

            def run():
                B.check(lib.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                                   B.ptr(out), n, n, ci, co, 27, 0, 1, None, None, 0, None, None,
                                                   B.stream()), 'conv')
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            ok = torch.equal(out, ref)
            res.append('%10.1f %s' % (timeit(run), 'ok ' if ok else 'DIFF'))
        dbg(0, 0)
        dbg(1, 0)
        print('s%-2d %3d->%-3d (%4dk rows)   ' % (stride, ci, co, n // 1000), ''.join('%16s' % r for r in res), flush=True)


if __name__ == '__main__':
    main()
