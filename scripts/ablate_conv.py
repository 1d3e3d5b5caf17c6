"""Timing ablation of lidal_conv_apply (cdna_hip_programming.md section 7 'Ablate').

  python scripts/ablate_conv.py build      # here (no GPU): one .so per ablation mask
  python scripts/ablate_conv.py run        # on the GPU box: time every variant on the bench layer
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'scripts', '_abl')
MASKS = {0: 'full', 1: 'no A gather', 2: 'no W staging', 4: 'no MFMA', 16: 'no epilogue',
         3: 'no A, no W', 7: 'no A/W/MFMA', 23: 'nothing (skeleton)'}


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'lidal_amd', 'csrc')
    for m in MASKS:
        lib = os.path.join(OUT, 'conv_%d.so' % m)
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-shared',
               '-DLIDAL_ABLATE=%d' % m, '-x', 'hip', os.path.join(csrc, 'conv.hip'), '-x', 'hip',
               os.path.join(csrc, 'error.cpp'), '-o', lib]
        subprocess.run(cmd, check=True)
        print('built', lib)


def run():
    import torch
    from lidal_amd import backend as B, synth
    from lidal_amd.nn import functional as F
    frames = int(os.environ.get('ABL_FRAMES', '5'))
    dtype = torch.bfloat16 if os.environ.get('ABL_DTYPE', 'bf16') == 'bf16' else torch.float32
    ci = int(os.environ.get('ABL_CI', '96'))
    co = int(os.environ.get('ABL_CO', '96'))
    batch = synth.make_train_batch(n_frames=frames, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).cuda()
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    n = coords.shape[0]
    x = torch.randn(n, ci, device='cuda').to(dtype)
    wk = (torch.randn(27, co, ci, device='cuda') * 0.02).to(dtype)
    out = torch.empty((n, co), dtype=dtype, device='cuda')
    sig = B.SIGNATURES['lidal_conv_apply']
    print('rows %d rules %d  ci %d co %d %s' % (n, kmap.total, ci, co, dtype))
    for m, name in MASKS.items():
        lib = ctypes.CDLL(os.path.join(OUT, 'conv_%d.so' % m))
        fn = lib.lidal_conv_apply
        fn.restype, fn.argtypes = sig

        def launch():
            rc = fn(B.ptr(x), B.ptr(wk), B.ptr(kmap.nbr_out), B.ptr(out), n, ci, co, 27, 0,
                    B.dtype_code(dtype), B.stream())
            assert rc == 0
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            launch()
        e1.record()
        torch.cuda.synchronize()
        print('%-22s %8.1f us' % (name, e0.elapsed_time(e1) * 100))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
