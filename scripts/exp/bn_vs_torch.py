"""GPU experiment: the element-wise BatchNorm launches against torch's plain vectorised kernels over the same bytes
(clamp_min: one array read, one written; add: two read, one written) -- how far the streaming parts are from what the
memory system gives."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B  # noqa: E402
from exp_img import timeit  # noqa: E402


def main():
    dev = torch.device('cuda')
    L = B.lib()
    print('lib', B.LIB_PATH)
    for n, c in ((396662, 96), (396662, 256), (396662, 32), (226469, 96), (105363, 128), (43145, 256)):
        x = torch.randn(n, c, device=dev).bfloat16()
        go = torch.randn(n, c, device=dev).bfloat16()
        y, dx = torch.empty_like(x), torch.empty_like(x)
        w, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        mean, invstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        gg, gb = torch.empty(c, device=dev), torch.empty(c, device=dev)
        tiles = -(-n // 128)
        ts = torch.zeros(c, tiles, 3, device=dev)
        ts[:, :, 0] = 128.0
        ts[:, :, 2] = 128.0
        sums = torch.zeros(c, tiles, 2, device=dev)

        def ev():
            B.check(L.lidal_bn_eval_fwd(B.ptr(x), 1, n, c, B.ptr(w), B.ptr(b), B.ptr(rm), B.ptr(rv), 1e-5, 1, B.ptr(y), B.stream()), 'eval')

        def ft():
            B.check(L.lidal_bn_train_fwd_tiles(B.ptr(x), 1, n, c, B.ptr(w), B.ptr(b), 1e-5, 0.1, B.ptr(rm), B.ptr(rv), None, 1,
                                               None, B.ptr(y), B.ptr(mean), B.ptr(invstd), B.ptr(ts), tiles, B.stream()), 'fwd')

        def bt():
            B.check(L.lidal_bn_bwd_tiles(B.ptr(x), B.ptr(go), c, 1, n, c, B.ptr(w), B.ptr(b), 1, B.ptr(mean), B.ptr(invstd),
                                         B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(sums), tiles, B.stream()), 'bwd')
        t_ev, t_ft, t_bt = timeit(ev), timeit(ft), timeit(bt)
        L.lidal_bn_set_fused(0)
        t_ft0, t_bt0 = timeit(ft), timeit(bt)
        L.lidal_bn_set_fused(1)
        t_clamp = timeit(lambda: torch.clamp_min(x, 0, out=y))
        t_add = timeit(lambda: torch.add(x, go, out=dx))
        mb = n * c * 2 / 1e6
        print('%7d x %3d (%6.1f MB)  eval_fwd %6.1f | fwd_tiles fused %6.1f apart %6.1f | torch clamp %6.1f || bwd_tiles fused %6.1f '
              'apart %6.1f | torch add %6.1f  us' % (n, c, mb, t_ev, t_ft, t_ft0, t_clamp, t_bt, t_bt0, t_add), flush=True)


main()
