"""GPU: the library's radix sort (csrc/sort.hip) against torch.sort (rocPRIM) on the list sizes of a
train step: us per sort of (u32 key, i32 value) pairs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B  # noqa: E402
from exp_img import timeit  # noqa: E402


def main():
    dev = 'cuda'
    print('%10s %5s %12s %14s' % ('items', 'bits', 'sort.hip us', 'torch.sort us'))
    for n, bits in ((16730, 27), (43145, 27), (105363, 27), (226469, 27), (396662, 27), (396662, 8), (226469, 8),
                    (396662, 19), (3173296, 19), (3173296, 15)):
        keys = torch.randint(0, 1 << bits, (n,), device=dev, dtype=torch.int32)
        vals = torch.arange(n, device=dev, dtype=torch.int32)
        ko, vo = torch.empty_like(keys), torch.empty_like(vals)
        nbytes = B.lib().lidal_sort_pairs_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        def ours():
            B.check(B.lib().lidal_sort_pairs(B.ptr(keys), B.ptr(vals), B.ptr(ko), B.ptr(vo), n, bits, B.ptr(ws),
                                             nbytes, B.stream()), 'sort')
        print('%10d %5d %12.1f %14.1f' % (n, bits, timeit(ours), timeit(lambda: torch.sort(keys, stable=True))),
              flush=True)


def main64():
    dev = 'cuda'
    print('%10s %5s %5s %12s %14s' % ('items', 'bits', 'vals', 'sort.hip us', 'torch.sort us'))
    for n, bits, with_vals in ((396662, 60, False), (180000, 48, False), (75000, 48, False), (28000, 48, False),
                               (120000, 62, True), (120000, 39, True)):
        keys = torch.randint(0, 1 << bits, (n,), device=dev, dtype=torch.int64)
        vals = torch.arange(n, device=dev, dtype=torch.int32)
        ko, vo = torch.empty_like(keys), torch.empty_like(vals)
        nbytes = B.lib().lidal_sort_pairs_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        def ours():
            B.check(B.lib().lidal_sort_pairs_u64(B.ptr(keys), B.ptr(vals) if with_vals else None, B.ptr(ko),
                                                 B.ptr(vo) if with_vals else None, n, bits, B.ptr(ws), nbytes,
                                                 B.stream()), 'sort')
        print('%10d %5d %5s %12.1f %14.1f' % (n, bits, with_vals, timeit(ours),
                                              timeit(lambda: torch.sort(keys, stable=with_vals))), flush=True)


def skewed():
    """the occupancy masks of a real level: few distinct keys, long runs"""
    dev = 'cuda'
    from lidal_amd import synth
    from lidal_amd.nn import functional as F
    b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(b['coords_v_b']).to(dev)
    km, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    nbr = km.nbr_out
    n = nbr.shape[1]
    from exp_img import timeit as t
    print('kmap_order of the level-0 3x3x3 table (%d rows): %.1f us' % (n, t(lambda: F.conv.RowOrder(nbr))))


if __name__ == '__main__':
    main64()
    skewed()
    main()
