"""Dense per-row products on [N, C] feature matrices: the 1x1x1 convolutions (F.conv3d's first
branch, `feats.matmul(weight)`) and the point-branch nn.Linear layers (network/spvcnn.py:85-101).

Forward and data gradient are plain library GEMMs, as upstream.  The WEIGHT gradient
x^T [Cin, N] @ g [N, Cout] reduces over N ~ 4e5 rows into a tiny [Cin, Cout] tile: a library GEMM
runs that in a few workgroups (0.8 ms measured for 128x96), so it goes through the split-K MFMA
kernel used for the sparse weight gradients (lidal_conv_wgrad with the identity rule list)."""
import torch
from torch.autograd import Function

from ... import backend as B
from .conv import WGRAD_CHUNK, _wgrad_splits

__all__ = ['rows_matmul', 'rows_linear']

_koff_cache = {}


def _koff(n, device):
    key = (n, str(device))
    t = _koff_cache.get(key)
    if t is None:
        if len(_koff_cache) > 256:
            _koff_cache.clear()
        t = torch.tensor([0, n], dtype=torch.int64, device=device)
        _koff_cache[key] = t
    return t


def _wgrad_dense(a, b):
    """a [N, Ca], b [N, Cb] (same dtype, f32 or bf16) -> a^T @ b as f32 [Ca, Cb]."""
    n, ca = a.shape
    cb = b.shape[1]
    splits = _wgrad_splits(n)
    gw = torch.empty((1, ca, cb), dtype=torch.float32, device=a.device)
    partial = torch.empty((splits, 1, ca, cb), dtype=torch.float32, device=a.device)
    B.check(B.lib().lidal_conv_wgrad(B.ptr(a), B.ptr(b), None, B.ptr(_koff(n, a.device)), 0,
                                     B.ptr(gw), B.ptr(partial), splits, WGRAD_CHUNK, 1, ca, cb,
                                     B.dtype_code(a.dtype), B.stream()), 'conv_wgrad(dense)')
    return gw[0]


def _ok(x, ca, cb):
    vec = 8 if x.dtype == torch.bfloat16 else 4
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and ca % vec == 0
            and cb % vec == 0 and x.shape[0] > 0)


class RowsMatmul(Function):
    """y = x @ w  (+ bias), w [Cin, Cout]."""

    @staticmethod
    def forward(ctx, x, w, bias):
        cdtype = torch.bfloat16 if torch.is_autocast_enabled() else x.dtype
        xc = x.contiguous().to(cdtype)
        wc = w.detach().to(cdtype)
        y = xc @ wc
        if bias is not None:
            y = y + bias.detach().to(cdtype)
        ctx.save_for_backward(xc, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        xc, w = ctx.saved_tensors
        g = g.contiguous().to(xc.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = g @ w.detach().to(xc.dtype).t()
        if ctx.needs_input_grad[1]:
            if _ok(xc, xc.shape[1], g.shape[1]):
                gw = _wgrad_dense(xc, g).to(w.dtype)
            else:
                gw = (xc.float().t() @ g.float()).to(w.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.float().sum(0)
        return gx, gw, gb


def rows_matmul(x, w, bias=None):
    return RowsMatmul.apply(x, w, bias)


def rows_linear(x, weight, bias=None):
    """nn.Linear semantics: weight [Cout, Cin]."""
    return RowsMatmul.apply(x, weight.t(), bias)
