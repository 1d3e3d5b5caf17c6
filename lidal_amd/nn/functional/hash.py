"""F.sphash (torchsparse/nn/functional/hash.py; network/utils.py:17,42-47,70-75)."""
import torch

from ... import backend as B

__all__ = ['sphash']


def sphash(coords, offsets=None):
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    B.require_gpu(coords)
    coords = coords.contiguous()
    n = coords.shape[0]
    if offsets is None:
        out = B.empty(n, torch.int64, coords.device)
        B.check(B.lib().lidal_hash(B.ptr(coords), n, B.ptr(out), B.stream()), 'hash')
        return out
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    B.require_gpu(offsets)
    offsets = offsets.contiguous()
    k = offsets.shape[0]
    out = B.empty((k, n), torch.int64, coords.device)
    B.check(B.lib().lidal_kernel_hash(B.ptr(coords), n, B.ptr(offsets), k, B.ptr(out),
                                      B.stream()), 'kernel_hash')
    return out
