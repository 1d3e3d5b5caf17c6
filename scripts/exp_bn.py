"""GPU experiment: achieved bandwidth of the BatchNorm kernels per (rows, channels) of the bench
batch's levels (algorithmic bytes: forward 3 N C b, backward 5 N C b)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B  # noqa: E402
from lidal_amd.nn.functional.norm import batch_norm_rows  # noqa: E402
from exp_img import timeit  # noqa: E402

SHAPES = [(396662, 32), (396662, 96), (226000, 32), (226000, 64), (105000, 64), (105000, 128), (43000, 128),
          (43000, 256), (16000, 256)]


def main():
    dev = torch.device('cuda')
    print('lib', B.LIB_PATH)
    tot_f = tot_b = 0.0
    for n, c in SHAPES:
        x = torch.randn(n, c, device=dev).bfloat16().requires_grad_(True)
        w = torch.ones(c, device=dev, requires_grad=True)
        b = torch.zeros(c, device=dev, requires_grad=True)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        go = torch.randn(n, c, device=dev).bfloat16()

        def fwd():
            return batch_norm_rows(x, w, b, rm, rv, True, 0.1, 1e-5, True)
        y = fwd()

        def bwd():
            torch.autograd.grad(y, (x, w, b), go, retain_graph=True)
        with torch.no_grad():
            tf = timeit(fwd)
        tb = timeit(bwd)
        by = n * c * 2
        tot_f += tf
        tot_b += tb
        print('%7d x %3d   fwd %6.1f us (%5.2f TB/s)   bwd %6.1f us (%5.2f TB/s)'
              % (n, c, tf, 3 * by / tf / 1e6, tb, 5 * by / tb / 1e6), flush=True)
    print('sum fwd %.1f us  bwd %.1f us' % (tot_f, tot_b))


if __name__ == '__main__':
    main()
