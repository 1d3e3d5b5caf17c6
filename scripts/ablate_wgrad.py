"""A/B timing of lidal_conv_wgrad variants in one process (build here, run on the GPU box)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'scripts', '_abl')
VARIANTS = {'base': []}
CHUNKS = [512, 1024, 2048, 4096, 8192]
SHAPES = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (4, 128, 128), (8, 256, 256), (8, 384, 256), (16, 256, 256)]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'lidal_amd', 'csrc')
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, 'wgrad_%s.so' % name)
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-shared'] + flags +
                       ['-x', 'hip', os.path.join(csrc, 'conv.hip'), '-x', 'hip',
                        os.path.join(csrc, 'error.cpp'), '-o', lib], check=True)
        print('built', lib)


def run():
    import torch
    from lidal_amd import backend as B, synth
    from lidal_amd.nn import functional as F
    from lidal_amd.nn.functional.conv import _wgrad_splits  # noqa: F401
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).cuda()
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    sig = B.SIGNATURES['lidal_conv_wgrad']
    fns = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, 'wgrad_%s.so' % name))
        lib.lidal_conv_wgrad.restype, lib.lidal_conv_wgrad.argtypes = sig
        fns[name] = lib.lidal_conv_wgrad
    print('%-26s' % 'shape' + ''.join('%10s' % ('chunk%d' % c) for c in CHUNKS))
    for stride, ci, co in SHAPES:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        x = torch.randn(n, ci, device='cuda').bfloat16()
        g = torch.randn(n, co, device='cuda').bfloat16()
        gw = torch.empty((27, ci, co), device='cuda')
        ts = []
        fn = fns['base']
        for chunk in CHUNKS:
            splits = max(1, min(256, -(-n // chunk)))
            part = torch.empty((splits, 27, ci, co), device='cuda')

            def launch():
                assert fn(B.ptr(x), B.ptr(g), B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0, B.ptr(gw),
                          B.ptr(part), splits, chunk, 27, ci, co, 1, B.stream()) == 0
            for _ in range(2):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 200)
        print('%-26s' % ('s%d %d->%d (%dk rows)' % (stride, ci, co, n // 1000)) + ''.join('%10.1f' % t for t in ts))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
