"""Timing experiments on lidal_conv_apply (cdna_hip_programming.md section 7 'Ablate').

  python scripts/ablate_conv.py build      # here (no GPU): one .so per variant
  python scripts/ablate_conv.py run        # on the GPU box: time every variant on the model's shapes
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'scripts', '_abl')
VARIANTS = {'base': [], 'rowb256': ['-DLIDAL_ROWB_OVERRIDE=256'], 'base2': []}
# (level stride, ci, co): the heavy layer families of the U-Net
SHAPES = [(1, 32, 32), (1, 96, 96), (2, 32, 32), (4, 128, 128), (4, 64, 64), (8, 256, 256),
          (8, 384, 256), (16, 256, 256)]


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'lidal_amd', 'csrc')
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, 'conv_%s.so' % name)
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-shared'] + flags + \
              ['-x', 'hip', os.path.join(csrc, 'conv.hip'), '-x', 'hip',
               os.path.join(csrc, 'error.cpp'), '-o', lib]
        subprocess.run(cmd, check=True)
        print('built', lib)


def run():
    import torch
    from lidal_amd import backend as B, synth
    from lidal_amd.nn import functional as F
    dtype = torch.bfloat16 if os.environ.get('ABL_DTYPE', 'bf16') == 'bf16' else torch.float32
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).cuda()
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    sig = B.SIGNATURES['lidal_conv_apply']
    libs = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, 'conv_%s.so' % name))
        lib.lidal_conv_apply.restype, lib.lidal_conv_apply.argtypes = sig
        libs[name] = lib.lidal_conv_apply
    print('%-22s' % 'shape (rows, rules)' + ''.join('%11s' % (n[:5] + '+sort') for n in VARIANTS))
    for stride, ci, co in SHAPES:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        x = torch.randn(n, ci, device='cuda').to(dtype)
        wk = (torch.randn(27, co, ci, device='cuda') * 0.02).to(dtype)
        out = torch.empty((n, co), dtype=dtype, device='cuda')
        row = 's%d %d->%d (%dk,%dk)' % (stride, ci, co, n // 1000, kmap.total // 1000)
        times = []
        stamps = ''
        order = kmap.order_out
        for name, fn in [(n_ + m_, f_) for n_, f_ in libs.items() for m_ in ('+sort',)]:
            tab, prm, tmk = ((order.table, order.perm, order.tile_masks) if name.endswith('+sort')
                             else (kmap.nbr_out, None, None))
            def launch():
                rc = fn(B.ptr(x), B.ptr(wk), B.ptr(tab), B.ptr(prm), B.ptr(tmk), B.ptr(out), n, n, ci, co, 27, 0,
                        B.dtype_code(dtype), None, None, 0, None, B.stream())
                assert rc == 0
            if fn(B.ptr(x), B.ptr(wk), B.ptr(tab), B.ptr(prm), B.ptr(tmk), B.ptr(out), n, n, ci, co, 27, 0,
                  B.dtype_code(dtype), None, None, 0, None, B.stream()) != 0:
                times.append(float('nan'))
                continue
            for _ in range(2):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                launch()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 200)
            if name.startswith('stamp'):
                lib = ctypes.CDLL(os.path.join(OUT, 'conv_stamp.so'))
                buf = (ctypes.c_ulonglong * 8)()
                lib.lidal_debug_stamps(buf, 1)
                tot = sum(buf[:4]) or 1
                stamps = '   [wave-phase cycles: issue %.0f  mfma %.0f  store %.0f  barrier %.0f | shares %s]' % (
                    buf[0] / max(buf[4], 1), buf[1] / max(buf[4], 1), buf[2] / max(buf[4], 1),
                    buf[3] / max(buf[4], 1), ' '.join('%.0f%%' % (100 * b / tot) for b in buf[:4]))
        print('%-22s' % row + ''.join('%11.1f' % t for t in times) + (stamps if 'stamp' in VARIANTS else ''))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
