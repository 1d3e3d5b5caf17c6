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


if __name__ == '__main__':
    main()
