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

__all__ = ['Conv3d', 'BatchNorm', 'BatchNorm1d', 'Linear', 'ReLU', 'functional', 'utils', 'Deferred']

# LIDAL_SURFACE_FUSION=0: every surface module computes its output when it is called (rounds 1-4)
import os as _os
SURFACE_FUSION = _os.environ.get('LIDAL_SURFACE_FUSION', '1') != '0'       # deferred BatchNorm (class Deferred)
ASSUME_BN_FOLLOWS = SURFACE_FUSION                                          # Conv3d leaves tile statistics (Conv3d.bn_follows)
# The coordinate tables of a whole forward pass at once, by LOOK-BACK (round 6; rounds 4-5 guessed: "the first 3x3x3 stride-1
# convolution on a fresh coordinate set is the stem of the reference's U-Net" -- a benchmark-shaped heuristic inside a generic
# operator).  Every forward pass records, on the map cache of its input, which (stride, kernel, conv stride) maps its
# convolutions ask for, in order; the Conv3d module that ran FIRST on that fresh coordinate set keeps the record.  When the same
# module meets the next fresh coordinate set it prefetches exactly what the previous pass used, if that was an encoder chain
# (prefetch_kernel_maps: one sort for the coarser levels, one chain of launches for the maps, one sort per kernel volume for
# the row orders -- instead of one by one as the convolutions ask for them, each with its own host round trips).  Same tables
# under the same keys; a module's first pass, and networks whose maps do not form a chain, build lazily as torchsparse does.
# LIDAL_SURFACE_PYRAMID=0: always lazily.
SURFACE_PYRAMID = _os.environ.get('LIDAL_SURFACE_PYRAMID', '1') != '0'


def _plan_from_trace(trace, stride):
    """The (kernel, conv stride) sequence prefetch_kernel_maps walks, if the recorded maps form a chain that starts at
    `stride` (every map's input stride is the stride reached by the strided maps before it); else None."""
    if not trace or not trace['maps']:
        return None
    plan, cur = [], tuple(stride)
    seen = {cur}
    for in_stride, kernel_size, conv_stride in trace['maps']:
        if tuple(in_stride) != cur:
            if tuple(in_stride) in seen:          # a level visited before (the decoder returning to it): its map is known
                continue
            return None
        plan.append((tuple(kernel_size), tuple(conv_stride)))
        if any(v > 1 for v in conv_stride):
            cur = tuple(cur[k] * conv_stride[k] for k in range(3))
            seen.add(cur)
    return tuple(plan)


class Deferred:
    """y = bn(x) of a train-mode spnn.BatchNorm, not computed yet.  What may join it before somebody reads the features:

        nn.Sequential(spnn.Conv3d, spnn.BatchNorm, spnn.ReLU(True))           -> relu=True       (network/utils.py:109-117)
        relu(net(x) + downsample(x)), net ending in spnn.BatchNorm             -> residual, relu_after  (utils.py:142-172)

    i.e. exactly the operands of norm.batch_norm_rows (lidal_bn_train_fwd[_tiles]: ReLU and the residual sum inside the
    normalising pass, the ReLU mask inside the backward kernels) that lidal_amd.network passes explicitly.  Same
    arithmetic, same roundings as the separate operators (the fused kernels round the summand they produce before
    adding), hence bitwise the undeferred results (tests/test_model_gpu.py).  One Deferred belongs to ONE SparseTensor;
    `plus` hands the pending normalisation over to the sum's tensor -- if the pre-sum tensor is read after all (nobody in
    the reference does), it is computed on its own and the sum falls back to a plain addition.
    Timing: the running statistics and num_batches_tracked of the module are updated when the features are first READ, not
    when the module is called -- and not at all if nobody ever reads them (a tensor that is dropped unread was never
    normalised; eager torch would have counted the batch)."""

    __slots__ = ('module', 'x', 'stats', 'relu', 'residual', 'relu_after', 'parent', 'moved', 'value')

    def __init__(self, module, x, stats):
        self.module, self.x, self.stats = module, x, stats
        self.relu = False           # an in-place ReLU directly behind the norm
        self.residual = None        # [N, C] added to the norm's output
        self.relu_after = False     # an in-place ReLU behind that sum
        self.parent = None          # plus(): the Deferred of the pre-sum tensor this one took over
        self.moved = None           # ... and, on that one, the sum's Deferred
        self.value = None

    def can_take_sum(self):
        return self.residual is None and not self.relu and self.moved is None and self.value is None

    def can_take_relu(self):
        return self.value is None and self.moved is None and not (self.relu_after if self.residual is not None else self.relu)

    def plus(self, other_feats):
        d = Deferred(self.module, self.x, self.stats)
        d.residual = other_feats
        d.parent = self
        self.moved = d
        return d

    def take_relu(self):
        if self.residual is not None:
            self.relu_after = True
        else:
            self.relu = True

    def _run(self, relu, residual, relu_after, momentum=None, count=True):
        from .functional import norm
        m = self.module
        return norm.batch_norm_rows(self.x, m.weight, m.bias, m.running_mean, m.running_var, True,
                                    m.momentum if momentum is None else momentum, m.eps, relu,
                                    m.num_batches_tracked if count else None, self.stats, residual, relu_after)

    def resolve(self):
        if self.value is not None:
            return self.value
        with torch.enable_grad():       # (deferred under autograd: a first read inside no_grad must not lose the graph)
            return self._resolve()

    def _resolve(self):
        if self.parent is not None and self.parent.value is not None:
            # (not reached since round 6: a pre-sum tensor that is read first fills its sum in the same breath, below)
            y = self.parent.value + self.residual
            self.value = torch.relu(y) if self.relu_after else y
        elif self.moved is not None and self.moved.value is not None:
            # the sum was computed (fused) first and now the pre-sum tensor is read: bn(x) once more, without touching
            # the running statistics a second time (momentum 0, no batch count)
            self.value = self._run(self.relu, None, False, momentum=0.0, count=False)
        elif self.moved is not None:
            # the pre-sum tensor is read BEFORE its sum: bn(x) on its own, and the sum -- a plain addition (and ReLU) now --
            # at once, while nobody can have written into bn(x) yet (an in-place operation on the pre-sum
            # tensor between the two reads would otherwise have leaked into the sum)
            self.value = self._run(self.relu, None, False)
            m = self.moved
            y = self.value + m.residual
            m.value = torch.relu(y) if m.relu_after else y
            m.x = m.stats = None
        else:
            self.value = self._run(self.relu, self.residual, self.relu_after)
        self.x = self.stats = None
        return self.value


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
        from .. import _note_conv3d
        _note_conv3d(self)          # (install_as_torchsparse: the model this layer ends up in is adopted at its first call)

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

    # Does a train-mode BatchNorm consume the output?  lidal_amd.network says so per layer (True / False); the surface
    # alone cannot know what follows in the user's nn.Sequential, and ASSUMES it (None): every Conv3d of the reference's
    # networks is followed by one (network/utils.py:109-117,127-136,147-168).  The convolution's epilogue then leaves the
    # per-tile (count, mean, M2) with its output; a BatchNorm that finds them skips its statistics pass, anything else
    # ignores them (the epilogue costs about what that pass costs, so a wrong guess costs one pass).
    bn_follows = None

    def forward(self, input, fork=False):
        """`fork` (k > 1, not transposed): returns (output, alias of `input`) for a second consumer of
        the input whose gradient then joins this layer's data gradient in-kernel (functional/conv.py)."""
        follows = ASSUME_BN_FOLLOWS if self.bn_follows is None else self.bn_follows
        if (SURFACE_PYRAMID and not self.transposed and self.kernel_volume > 1 and not input.kmaps
                and getattr(input.kmaps, 'trace', None) is None and input.coords.is_cuda and input.coords.shape[0] > 0):
            # the first convolution on a fresh coordinate set: prefetch what the previous pass that began here used
            prev = self.__dict__.get('_lidal_trace')
            plan = _plan_from_trace(prev, input.stride)
            input.kmaps.trace = self.__dict__['_lidal_trace'] = {'maps': [], 'transposed': False}
            if plan:
                from .functional.conv import prefetch_kernel_maps
                prefetch_kernel_maps(input, plan, transposed=prev['transposed'])
        out = conv3d(input, self.kernel, kernel_size=self.kernel_size, bias=self.bias,
                     stride=self.stride, dilation=self.dilation, transposed=self.transposed,
                     want_stats=follows and self.training and torch.is_grad_enabled(), fork=fork)
        f = (out[0] if fork else out)._feats
        if getattr(f, '_lidal_bn_stats', None) is not None:
            f._lidal_bn_stats_version = f._version      # an in-place write between here and the BatchNorm voids them
        return out


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
                                    _tile_stats_of(feats), residual, relu_after)


def _tile_stats_of(feats):
    """The tile statistics the producing convolution left with `feats`, unless the tensor was written since."""
    st = getattr(feats, '_lidal_bn_stats', None)
    if st is not None and getattr(feats, '_lidal_bn_stats_version', feats._version) != feats._version:
        return None
    return st


class BatchNorm(BatchNorm1d):
    """spnn.BatchNorm: BatchNorm1d applied to the .feats of a SparseTensor.  In training the result is DEFERRED
    (class Deferred) until it is read, so that a following in-place spnn.ReLU / residual sum joins the kernel."""

    def forward(self, input, residual=None, relu_after=False):
        feats = input.feats
        from .functional import norm
        if (SURFACE_FUSION and self.training and residual is None and not self.fused_relu and torch.is_grad_enabled()
                and self.track_running_stats and self.momentum is not None
                and norm.supported(feats, self.weight, self.bias)):
            out = SparseTensor(None, input.coords, input.stride)
            out._deferred = Deferred(self, feats, _tile_stats_of(feats))
            out.cmaps = input.cmaps
            out.kmaps = input.kmaps
            return out
        return fapply(input, super().forward, residual, relu_after)


class ReLU(nn.ReLU):
    def forward(self, input):
        d = input._deferred
        if d is not None and self.inplace and d.can_take_relu():
            d.take_relu()               # in place: `input` itself now stands for relu(...)
            return input
        return fapply(input, super().forward)
