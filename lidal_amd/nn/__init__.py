"""torchsparse.nn modules used by the reference (network/utils.py:109-117, network/spvcnn.py:21-25):
Conv3d (parameter `kernel` [K, Cin, Cout], the checkpoint compatibility surface), BatchNorm
(an nn.BatchNorm1d over .feats), ReLU."""
import math

import numpy as np
import torch
from torch import nn

from .. import backend as B
from ..tensor import SparseTensor
from ..utils import make_ntuple
from . import functional, utils
from .functional import conv3d

__all__ = ['Conv3d', 'BatchNorm', 'BatchNorm1d', 'Linear', 'ReLU', 'functional', 'utils']


def fapply(input, fn, *args, **kwargs):
    out = SparseTensor(fn(input.feats, *args, **kwargs), input.coords, input.stride)
    out.cmaps = input.cmaps
    out.kmaps = input.kmaps
    return out


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1,
                 bias=False, transposed=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = make_ntuple(kernel_size, ndim=3)
        self.stride = make_ntuple(stride, ndim=3)
        self.dilation = dilation
        self.transposed = transposed
        self.kernel_volume = int(np.prod(self.kernel_size))
        shape = ((self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1
                 else (in_channels, out_channels))
        self.kernel = nn.Parameter(torch.zeros(*shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def extra_repr(self):
        s = '{in_channels}, {out_channels}, kernel_size={kernel_size}'
        if self.stride != (1,) * 3:
            s += ', stride={stride}'
        if self.transposed:
            s += ', transposed=True'
        return s.format(**self.__dict__)

    def reset_parameters(self):
        fan = (self.out_channels if self.transposed else self.in_channels) * self.kernel_volume
        std = 1 / math.sqrt(fan)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    bn_follows = False      # set by lidal_amd.network where a BatchNorm directly consumes the output

    def forward(self, input, fork=False):
        """`fork` (k > 1, not transposed): returns (output, alias of `input`) for a second consumer of
        the input whose gradient then joins this layer's data gradient in-kernel (functional/conv.py)."""
        return conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias,
                      stride=self.stride, dilation=self.dilation, transposed=self.transposed,
                      want_stats=self.bn_follows and self.training and torch.is_grad_enabled(), fork=fork)


class Linear(nn.Linear):
    """nn.Linear on a [N, C] tensor whose weight gradient runs on the split-K MFMA kernel (see
    functional/dense.py); parameters are nn.Linear's, so state_dict keys are unchanged."""

    bn_follows = False

    def forward(self, x):
        if not x.is_cuda or x.dim() != 2:
            B.hit('torch_fallback:Linear')
            return super().forward(x)
        from .functional.dense import rows_linear
        return rows_linear(x, self.weight, self.bias,
                           want_stats=self.bn_follows and self.training and torch.is_grad_enabled())


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d on a [N, C] tensor, routed to the HIP kernels (lidal_bn_*) whenever the
    configuration is the standard one (GPU, f32/bf16, affine, running statistics); anything else
    falls through to torch's own implementation.  Parameters / buffers are nn.BatchNorm1d's, so
    state_dict keys are unchanged."""

    fused_relu = False      # set by lidal_amd.network where a ReLU directly follows the norm

    def forward(self, feats, residual=None, relu_after=False):
        """`residual` ([N, C]): returns norm(feats) (+ ReLU) + residual (ReLU'd if `relu_after`); in
        training the sum happens inside the normalising pass."""
        from .functional import norm
        if (not norm.supported(feats, self.weight, self.bias) or not self.track_running_stats
                or self.momentum is None):
            B.hit('torch_fallback:BatchNorm1d')
            out = super().forward(feats)
            out = torch.relu(out) if self.fused_relu else out
            if residual is None:
                return out
            return torch.relu(out + residual) if relu_after else out + residual
        return norm.batch_norm_rows(feats, self.weight, self.bias, self.running_mean,
                                    self.running_var, self.training, self.momentum, self.eps,
                                    self.fused_relu, self.num_batches_tracked,
                                    getattr(feats, '_lidal_bn_stats', None), residual, relu_after)


class BatchNorm(BatchNorm1d):
    """spnn.BatchNorm: BatchNorm1d applied to the .feats of a SparseTensor."""

    def forward(self, input, residual=None, relu_after=False):
        return fapply(input, super().forward, residual, relu_after)


class ReLU(nn.ReLU):
    def forward(self, input):
        return fapply(input, super().forward)
