"""GPU: the lean convolution kernel against its deep form (slabs and gathers two phases ahead) on every layer
shape of the U-Net, bench batch (FRAMES scans): bitwise equality and time.  RECORD: it drove a temporary
lidal_debug_set_deep_rows(rows) hook and a launcher that offered the deep kernel for every tiling (commit
"conv_lean_deep_kernel: slabs and gathers two phases ahead ..."); results in profiles/README.md."""
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
    setrows = lib.lidal_debug_set_deep_rows
    setrows.argtypes = [ctypes.c_int64]
    batch = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '5')), n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    shapes = [(1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 128, 96), (2, 32, 32), (4, 64, 64), (4, 128, 128), (4, 192, 128),
              (4, 32, 64), (8, 128, 128), (8, 256, 256), (8, 384, 256), (8, 64, 128), (16, 256, 256), (16, 128, 256),
              (16, 256, 128)]
    print('%-28s %10s %10s %8s' % ('layer', 'lean us', 'deep us', 'equal'))
    for stride, ci, co in shapes:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
        x = torch.randn(n, ci, generator=g).to(dev).bfloat16()
        w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
        o = kmap.order_out
        img = make_image(w, torch.bfloat16, n)
        outs, ts = [], []
        for rows in (0, 1 << 40):
            setrows(rows)
            out = torch.empty((n, co), dtype=torch.bfloat16, device=dev)

            def run():
                B.check(lib.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                                   B.ptr(out), n, n, ci, co, 27, 0, 1, None, None, 0, None, None,
                                                   B.stream()), 'conv')
            run()
            torch.cuda.synchronize()
            outs.append(out.clone())
            ts.append(timeit(run))
        print('s%-2d %3d->%-3d (%4dk rows)      %10.1f %10.1f %8s' % (stride, ci, co, n // 1000, ts[0], ts[1],
                                                                     torch.equal(outs[0], outs[1])), flush=True)
    setrows(150000)


if __name__ == '__main__':
    main()
