"""torchsparse/nn/utils/kernel.py (v1.4.0) -- get_kernel_offsets.  TEST INFRASTRUCTURE.

Reference call site: network/utils.py:69 `get_kernel_offsets(2, x.s, 1, device=...)`.
"""
import numpy as np
import torch

from ..utils import make_ntuple

__all__ = ['get_kernel_offsets']


def get_kernel_offsets(size, stride=1, dilation=1, device='cpu'):
    size = make_ntuple(size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)
    axes = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k]
            for k in range(3)]
    # odd volume: x fastest (MinkowskiEngine weight order); even volume: z fastest
    if np.prod(size) % 2 == 1:
        offsets = [[x, y, z] for z in axes[2] for y in axes[1] for x in axes[0]]
    else:
        offsets = [[x, y, z] for x in axes[0] for y in axes[1] for z in axes[2]]
    return torch.tensor(np.asarray(offsets), dtype=torch.int, device=device)
