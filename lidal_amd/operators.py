"""torchsparse.cat (operators.py): channel concat of row-aligned sparse tensors
(network/spvcnn.py:133,137,145,149)."""
import torch

from .tensor import SparseTensor

__all__ = ['cat']


def cat(inputs):
    out = SparseTensor(torch.cat([x.feats for x in inputs], dim=1), inputs[0].coords,
                       inputs[0].stride)
    out.cmaps = inputs[0].cmaps
    out.kmaps = inputs[0].kmaps
    return out
