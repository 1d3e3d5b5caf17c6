"""What does the in-launch merge of the tile statistics cost?  lidal_bn_train_fwd_tiles (merge + apply in one launch) against
lidal_bn_eval_fwd (the same normalising pass with given statistics: the floor) and lidal_bn_bwd_tiles / lidal_bn_bwd on the
layer shapes of the bench batch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

dev = 'cuda'
L = B.lib()
print('%-22s %12s %12s %12s' % ('rows x channels', 'tiles+apply', 'apply only', 'merge share'))
for n, c in [(396662, 96), (396662, 32), (226469, 32), (105363, 64), (105363, 128), (43145, 256), (16730, 256), (600000, 256)]:
    x = torch.randn(n, c, device=dev).bfloat16()
    y = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    rm = torch.zeros(c, device=dev); rv = torch.ones(c, device=dev); nb = torch.zeros((), dtype=torch.int64, device=dev)
    mean = torch.empty(c, device=dev); inv = torch.empty(c, device=dev)
    tiles = -(-n // B.stats_tile_rows())
    st = torch.zeros(tiles, c, 3, device=dev)
    st[:, :, 0] = 128.0
    st[-1, :, 0] = n - 128.0 * (tiles - 1)
    st[:, :, 1] = 0.01
    st[:, :, 2] = 100.0
    def fused():
        B.check(L.lidal_bn_train_fwd_tiles(B.ptr(x), 1, n, c, B.ptr(g), B.ptr(b), 1e-5, 0.1, B.ptr(rm), B.ptr(rv), B.ptr(nb), 1,
                                           None, B.ptr(y), B.ptr(mean), B.ptr(inv), B.ptr(st), tiles, B.stream()), 'bn')
    def plain():
        B.check(L.lidal_bn_eval_fwd(B.ptr(x), 1, n, c, B.ptr(g), B.ptr(b), B.ptr(rm), B.ptr(rv), 1e-5, 1, B.ptr(y), B.stream()), 'bn')
    tf, tp = timeit(fused), timeit(plain)
    print('%8d x %-4d        %10.1f us %10.1f us %10.1f us' % (n, c, tf, tp, tf - tp))
