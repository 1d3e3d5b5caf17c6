"""Fused row-wise ops of the training step on the HIP backend: relu(a + b) of the residual blocks
(network/utils.py:171) and cross-entropy with ignore_index / mean reduction (train.py:136)."""
import torch
from torch.autograd import Function

from ... import backend as B

__all__ = ['add_relu', 'cross_entropy']


def _vec(dt):
    return 8 if dt == torch.bfloat16 else 4


def _add_relu_fwd(a, b):
    a, b = a.contiguous(), b.contiguous().to(a.dtype)
    y = torch.empty_like(a)
    B.check(B.lib().lidal_add_relu_fwd(B.ptr(a), B.ptr(b), B.ptr(y), a.numel(),
                                       B.dtype_code(a.dtype), B.stream()), 'add_relu_fwd')
    return y


class AddReLU(Function):
    @staticmethod
    def forward(ctx, a, b):
        y = _add_relu_fwd(a, b)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous().to(y.dtype)
        gin = torch.empty_like(y)
        B.check(B.lib().lidal_add_relu_bwd(B.ptr(y), B.ptr(g), B.ptr(gin), y.numel(),
                                           B.dtype_code(y.dtype), B.stream()), 'add_relu_bwd')
        return gin, gin


def add_relu(a, b):
    """relu(a + b) for two [N, C] feature matrices of the same shape."""
    if (a.is_cuda and a.shape == b.shape and a.dtype in (torch.float32, torch.bfloat16)
            and a.numel() % _vec(a.dtype) == 0 and a.numel() > 0):
        return AddReLU.apply(a, b) if B.wants_grad(a, b) else _add_relu_fwd(a, b)
    return torch.relu(a + b)


class CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits = logits.contiguous()
        labels = labels.contiguous()
        n, c = logits.shape
        out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
        nbytes = B.lib().lidal_ce_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=logits.device)
        B.check(B.lib().lidal_ce_fwd(B.ptr(logits), B.dtype_code(logits.dtype), B.ptr(labels), n, c,
                                     int(ignore_index), B.ptr(out2), B.ptr(ws), nbytes, B.stream()),
                'ce_fwd')
        ctx.save_for_backward(logits, labels, out2)
        ctx.ignore_index = int(ignore_index)
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        logits, labels, out2 = ctx.saved_tensors
        n, c = logits.shape
        d = torch.empty_like(logits)
        gs = g.reshape(1).float().contiguous()
        B.check(B.lib().lidal_ce_bwd(B.ptr(logits), B.dtype_code(logits.dtype), B.ptr(labels), n, c,
                                     ctx.ignore_index, B.ptr(out2), B.ptr(gs), B.ptr(d), B.stream()),
                'ce_bwd')
        return d, None, None


def cross_entropy(logits, labels, ignore_index=255):
    """torch.nn.functional.cross_entropy(logits, labels, ignore_index=..., reduction='mean')."""
    if (logits.is_cuda and logits.dim() == 2 and logits.shape[1] <= 32 and labels.dtype == torch.int64
            and logits.dtype in (torch.float32, torch.bfloat16) and logits.shape[0] > 0):
        return CrossEntropy.apply(logits, labels, ignore_index)
    return torch.nn.functional.cross_entropy(logits, labels, ignore_index=ignore_index,
                                             reduction='mean')
