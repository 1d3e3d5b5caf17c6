"""One training iteration, the counterpart of /root/reference/train.py:127-140:
zero_grad -> model(SparseTensor) -> cross_entropy(ignore_index=255, mean) -> backward -> Adam.step.
"""
import torch

from . import SparseTensor
from .nn.functional.fused import cross_entropy

__all__ = ['train_step', 'forward_backward']


def forward_backward(model, feats_v_b, coords_v_b, labels_v_b, autocast=False, geometry=None):
    x = SparseTensor(feats_v_b, coords_v_b)
    if geometry is not None:        # the batch's coordinate tables, built ahead (lidal_amd.network.GeometryPrefetcher)
        x.geometry = geometry
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        logits, _ = model(x)
    loss = cross_entropy(logits, labels_v_b, ignore_index=255)
    loss.backward()
    return loss, logits


def train_step(model, optimizer, feats_v_b, coords_v_b, labels_v_b, autocast=False, geometry=None):
    optimizer.zero_grad()
    loss, logits = forward_backward(model, feats_v_b, coords_v_b, labels_v_b, autocast, geometry)
    optimizer.step()
    return loss.detach(), logits.detach()
