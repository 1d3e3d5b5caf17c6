"""torchsparse.nn.utils.get_kernel_offsets (nn/utils/kernel.py v1.4.0; network/utils.py:69).

Host-side: K x 3 int32 offsets, cached per (size, stride, dilation, device) because the model
asks for the same handful of tables every step.
"""
import numpy as np
import torch

from ..utils import make_ntuple

__all__ = ['get_kernel_offsets']

_cache = {}


def get_kernel_offsets(size, stride=1, dilation=1, device='cpu'):
    size = make_ntuple(size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    key = (size, stride, dilation, str(device))
    hit = _cache.get(key)
    if hit is not None:
        return hit
    axes = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k]
            for k in range(3)]
    if np.prod(size) % 2 == 1:      # odd volume: x fastest (MinkowskiEngine weight order)
        grid = np.stack(np.meshgrid(axes[2], axes[1], axes[0], indexing='ij'), -1)
        offsets = grid.reshape(-1, 3)[:, ::-1]
    else:                           # even volume: z fastest
        grid = np.stack(np.meshgrid(axes[0], axes[1], axes[2], indexing='ij'), -1)
        offsets = grid.reshape(-1, 3)
    out = torch.tensor(np.ascontiguousarray(offsets), dtype=torch.int, device=device)
    _cache[key] = out
    return out
