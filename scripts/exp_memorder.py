"""GPU experiment: how much do the level-0 kernels depend on the MEMORY order of the rows?
SPVCNN's level 0 is in sorted-hash order (random in space, network/utils.py:18), MinkUNet's in the
dataset's lexicographic order.  Times conv_apply (shipped global mask sort), the weight gradient
and devoxelize-style 8-corner gathers are not included -- just the two MFMA kernels -- for both."""
import os
import sys

import numpy as np
import torch


def _gen1():
    """tests/native/liblidal_gen1.so: the first-generation kernel left the product library in round 4."""
    import ctypes, os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'native', 'liblidal_gen1.so'))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.lidal_conv_apply.restype = i32
    lib.lidal_conv_apply.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    return lib


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402
from lidal_amd.nn.functional import conv as C  # noqa: E402
from exp_img import timeit  # noqa: E402


def main():
    dev = torch.device('cuda')
    dtype = torch.bfloat16
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    base = batch['coords_v_b']
    for order in os.environ.get('LIDAL_EXP_ORDERS', 'dataset,hash').split(','):
        c_np = base if order == 'dataset' else base[np.random.default_rng(0).permutation(len(base))]
        coords = torch.from_numpy(np.ascontiguousarray(c_np)).to(dev)
        with torch.enable_grad():
            kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
        n = coords.shape[0]
        for ci, co in ((96, 96), (32, 32)):
            g = torch.Generator(device='cpu').manual_seed(1)
            x = torch.randn(n, ci, generator=g).to(dev).to(dtype)
            go = torch.randn(n, co, generator=g).to(dev).to(dtype)
            wt = (torch.randn(27, co, ci, generator=g) * 0.05).to(dev).to(dtype)
            o = kmap.order_out
            out = torch.empty((n, co), dtype=dtype, device=dev)

            def conv():
                B.check(_gen1().lidal_conv_apply(B.ptr(x), B.ptr(wt), B.ptr(o.table), B.ptr(o.perm),
                                                 B.ptr(o.tile_masks), B.ptr(out), n, n, ci, co, 27, 0,
                                                 B.dtype_code(dtype), None, None, 0, None, B.stream()), 'conv')
            gw = torch.empty((27, ci, co), dtype=torch.float32, device=dev)
            partial = C.wgrad_scratch(n, n, 27, ci, co, dtype, dev)

            def wgrad():
                B.check(B.lib().lidal_conv_wgrad(B.ptr(x), B.ptr(go), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0,
                                                 B.ptr(gw), B.ptr(partial), partial.shape[0], 27, ci, co,
                                                 B.dtype_code(dtype), B.stream()), 'wgrad')
            print('%-8s %3d->%-3d  conv_apply %7.1f us   wgrad %7.1f us (%d slabs)'
                  % (order, ci, co, timeit(conv), timeit(wgrad), partial.shape[0]), flush=True)


if __name__ == '__main__':
    main()
