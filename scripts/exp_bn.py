"""GPU experiment: achieved bandwidth of the BatchNorm launches per (rows, channels) of the bench
batch's levels, called straight through the C-ABI (no autograd): forward with the statistics pass
(3 N C b algorithmic bytes), forward on the convolution's tile statistics (2 N C b), backward
(5 N C b)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B  # noqa: E402
from exp_img import timeit  # noqa: E402

SHAPES = [(396662, 32), (396662, 96), (226000, 32), (226000, 64), (105000, 64), (105000, 128), (43000, 128),
          (43000, 256), (16000, 256)]
if os.environ.get('BN_SHAPES'):         # e.g. BN_SHAPES=396662x96 for a counter pass over one shape
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['BN_SHAPES'].split(',')]


def main():
    dev = torch.device('cuda')
    print('lib', B.LIB_PATH)
    L = B.lib()
    sweep(dev, L)


def sweep(dev, L):
    tot = [0.0, 0.0, 0.0]
    for n, c in SHAPES:
        x = torch.randn(n, c, device=dev).bfloat16()
        go = torch.randn(n, c, device=dev).bfloat16()
        y, dx = torch.empty_like(x), torch.empty_like(x)
        w, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        mean, invstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        gg, gb = torch.empty(c, device=dev), torch.empty(c, device=dev)
        nbytes = L.lidal_bn_workspace_bytes(n, c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        n_tiles = -(-n // 128)
        ts = torch.zeros(n_tiles, c, 3, device=dev)
        ts[:, :, 0] = 128.0
        ts[:, :, 2] = 128.0

        def fwd():
            B.check(L.lidal_bn_train_fwd(B.ptr(x), 1, n, c, B.ptr(w), B.ptr(b), 1e-5, 0.1, B.ptr(rm), B.ptr(rv),
                                         None, 1, None, B.ptr(y), B.ptr(mean), B.ptr(invstd), B.ptr(ws), nbytes,
                                         B.stream()), 'fwd')

        def fwd_tiles():
            B.check(L.lidal_bn_train_fwd_tiles(B.ptr(x), 1, n, c, B.ptr(w), B.ptr(b), 1e-5, 0.1, B.ptr(rm),
                                               B.ptr(rv), None, 1, None, B.ptr(y), B.ptr(mean), B.ptr(invstd),
                                               B.ptr(ts), n_tiles, B.stream()), 'fwd_tiles')

        def bwd():
            B.check(L.lidal_bn_bwd(B.ptr(x), B.ptr(go), c, 1, n, c, B.ptr(w), B.ptr(b), 1, B.ptr(mean),
                                   B.ptr(invstd), B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(ws), nbytes,
                                   B.stream()), 'bwd')
        def bwd_params():       # statistics of the backward only (dx = NULL): the reducing pass + its final
            B.check(L.lidal_bn_bwd(B.ptr(x), B.ptr(go), c, 1, n, c, B.ptr(w), B.ptr(b), 1, B.ptr(mean),
                                   B.ptr(invstd), None, B.ptr(gg), B.ptr(gb), B.ptr(ws), nbytes,
                                   B.stream()), 'bwd')
        fwd()
        t = [timeit(fwd), timeit(fwd_tiles), timeit(bwd)]
        tp = timeit(bwd_params)
        by = n * c * 2
        tot = [a + b_ for a, b_ in zip(tot, t)]
        print('%7d x %3d   fwd %6.1f us (%5.2f TB/s)   fwd on tile stats %6.1f us (%5.2f TB/s)   bwd %6.1f us (%5.2f TB/s)'
              % (n, c, t[0], 3 * by / t[0] / 1e6, t[1], 2 * by / t[1] / 1e6, t[2], 5 * by / t[2] / 1e6)
              + '   [reduce %5.1f us (%4.2f TB/s), dx %5.1f us (%4.2f TB/s)]'
              % (tp, 2 * by / tp / 1e6, t[2] - tp, 3 * by / (t[2] - tp) / 1e6), flush=True)
    print('sum fwd %.1f  fwd_tiles %.1f  bwd %.1f us' % tuple(tot))


if __name__ == '__main__':
    main()
