"""A/B timing of lidal_conv_wgrad builds in one process, over the layer shapes of the model.
Variants are the libraries scripts/build_variant.py left in scripts/_abl (`lib_<name>.so`, e.g.
built with -DLIDAL_WGRAD_RESIDENT=3 or -DLIDAL_WGRAD_NO_DMA=1); `shipped` is lidal_amd/liblidal_amd.so.
usage: ablate_wgrad.py shipped res3 nodma ..."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'scripts', '_abl')
SHAPES = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 64, 64), (4, 128, 128), (8, 256, 256), (8, 384, 256),
          (16, 256, 256)]


def run(names):
    import torch
    from lidal_amd import backend as B, synth
    from lidal_amd.nn import functional as F
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).cuda()
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    fns = {}
    for name in names:
        path = os.path.join(ROOT, 'lidal_amd', 'liblidal_amd.so') if name == 'shipped' \
            else os.path.join(OUT, 'lib_%s.so' % name)
        lib = ctypes.CDLL(path)
        for f in ('lidal_conv_wgrad', 'lidal_conv_wgrad_slabs'):
            getattr(lib, f).restype, getattr(lib, f).argtypes = B.SIGNATURES[f]
        fns[name] = lib
    print('%-28s' % 'shape' + ''.join('%16s' % n for n in names), flush=True)
    for stride, ci, co in SHAPES:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        x = torch.randn(n, ci, device='cuda').bfloat16()
        g = torch.randn(n, co, device='cuda').bfloat16()
        gw = torch.empty((27, ci, co), device='cuda')
        cells = []
        for name, lib in fns.items():
            slabs = lib.lidal_conv_wgrad_slabs(n, n, 27, ci, co, 1)
            part = torch.empty((slabs, ci, co), device='cuda')

            def launch():
                assert lib.lidal_conv_wgrad(B.ptr(x), B.ptr(g), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0,
                                            B.ptr(gw), B.ptr(part), slabs, 27, ci, co, 1, B.stream()) == 0
            for _ in range(2):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                launch()
            e1.record()
            torch.cuda.synchronize()
            cells.append('%8.1f (%4d)' % (e0.elapsed_time(e1) * 200, slabs))
        print('%-28s' % ('s%d %d->%d (%dk rows)' % (stride, ci, co, n // 1000)) + ''.join('%16s' % t for t in cells),
              flush=True)


if __name__ == '__main__':
    run(sys.argv[1:] or ['shipped'])
