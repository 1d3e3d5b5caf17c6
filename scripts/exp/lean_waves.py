"""GPU experiment: the lean convolution kernel on every layer shape of the bench batch -- time and a checksum of the
output -- for A/B builds of the library (LIDAL_AMD_LIB; e.g. scripts/build_variant.py lw16 conv_img.hip
-DLIDAL_LEAN_WAVES=16: 256-row tiles).  The checksums of two builds must agree (bit-equal outputs)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from exp_img import make_image, timeit  # noqa: E402
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402

SHAPES = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 32, 32), (2, 64, 64), (2, 96, 96), (2, 192, 96), (4, 64, 64),
          (4, 128, 128), (4, 256, 128), (8, 256, 256), (8, 384, 256), (16, 256, 256)]


def main():
    dev = torch.device('cuda')
    lib = B.lib()
    batch = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '5')), n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    print('library', B.LIB_PATH)
    for stride, ci, co in SHAPES:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
        x = torch.randn(n, ci, generator=g).to(dev).bfloat16()
        w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
        o = kmap.order_out
        out = torch.empty((n, co), dtype=torch.bfloat16, device=dev)
        img = make_image(w, torch.bfloat16, n)

        def run():
            B.check(lib.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                               B.ptr(out), n, n, ci, co, 27, 0, 1, None, None, 0, None, None, B.stream()), 'conv')
        run()
        torch.cuda.synchronize()
        h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        print('s%-2d %3d->%-3d (%4dk rows)  %8.1f us  %s' % (stride, ci, co, n // 1000, timeit(run), h), flush=True)


main()
