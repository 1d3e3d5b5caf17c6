"""F.spcount (torchsparse/nn/functional/count.py; network/utils.py:20,49)."""
import torch

from ... import backend as B

__all__ = ['spcount']


def spcount(coords, num):
    B.require_gpu(coords)
    idx = coords.contiguous()
    if idx.dtype != torch.int:
        idx = idx.int()
    out = B.empty(num, torch.int, idx.device)
    B.check(B.lib().lidal_count(B.ptr(idx), idx.numel(), B.ptr(out), num, B.stream()), 'count')
    return out
