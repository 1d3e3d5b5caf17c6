"""Building blocks of the sparse U-Net (counterparts of /root/reference/network/utils.py:105-172).

Attribute names (`net`, `downsample`, `relu`) and Sequential positions are the checkpoint
surface: state_dict keys such as `stage2.1.net.3.kernel` or `up1.1.0.downsample.1.weight` must
match the reference's so its checkpoints load with strict=True (train.py:66, prob_inference.py:68).
"""
import torch
from torch import nn

from .. import SparseTensor
from .. import backend as B
from .. import nn as spnn
from ..nn.functional.conv import conv3d
from ..nn.functional.dense import rows_linear
from ..nn.functional.fused import add_relu


class ConvNormSequential(nn.Sequential):
    """nn.Sequential (same child names, same state_dict keys) that, at inference (eval mode, no
    gradient wanted), runs each Conv3d -> BatchNorm (or Linear -> BatchNorm1d on a plain [N, C]
    tensor) pair as ONE kernel: the eval-mode BatchNorm is a per-channel affine map, applied (with
    the fused ReLU) in the convolution's epilogue instead of a separate pass over the feature
    matrix.  Anything else runs child by child."""

    def forward(self, x, residual=None, relu_after=False, start=0):
        """`residual` ([N, C] features): returns self(x) + residual (ReLU'd if `relu_after`); on the
        fused inference path the sum happens in the last layer's epilogue.  `start`: x is already the
        output of child start - 1."""
        mods = list(self)
        while mods and isinstance(mods[-1], nn.Identity):
            mods.pop()
        i = start
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, spnn.Conv3d) and isinstance(nxt, spnn.BatchNorm) and _fusable(m, nxt, x):
                last = i + 2 == len(mods) and residual is not None
                x = _conv_norm(m, nxt, x, residual if last else None, relu_after)
                if last:
                    residual = None
                i += 2
            elif (isinstance(m, spnn.Linear) and isinstance(nxt, spnn.BatchNorm1d)
                  and torch.is_tensor(x) and _fusable_rows(nxt, x, m.weight, m.bias)):
                scale, shift = _fold(nxt, x.device)
                last = i + 2 == len(mods)
                relu = int(nxt.fused_relu) | (2 if (last and relu_after and residual is not None) else 0)
                x = rows_linear(x, m.weight, m.bias,
                                epilogue=(scale, shift, relu, residual if last else None))
                if last:
                    residual = None
                i += 2
            elif (isinstance(m, spnn.BatchNorm1d) and not isinstance(m, spnn.BatchNorm) and torch.is_tensor(x)
                  and i + 1 == len(mods) and residual is not None and not relu_after and m.training
                  and (B.FORK & 4)):
                x = m(x, residual=residual)         # training: the sum rides in the normalising pass
                residual = None
                i += 1
            elif (isinstance(m, spnn.BatchNorm) and not torch.is_tensor(x) and i + 1 == len(mods)
                  and residual is not None and relu_after and m.training and (B.FORK & 8)):
                x = m(x, residual=residual, relu_after=True)    # training: relu(bn(x) + shortcut) in the normalising pass
                residual = None
                i += 1
            else:
                x = m(x)
                i += 1
        if residual is None:
            return x
        if torch.is_tensor(x):
            return add_relu(x, residual) if relu_after else residual + x
        out = SparseTensor(add_relu(x.F, residual) if relu_after else x.F + residual, x.C, x.s)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out


def _fusable_rows(bn, feats, *params):
    return (not bn.training and bn.track_running_stats and feats.is_cuda
            and not B.wants_grad(feats, bn.weight, bn.bias, *params)
            and bn.weight is not None and bn.weight.dtype == torch.float32)


def _fusable(conv, bn, x):
    return conv.bias is None and _fusable_rows(bn, x.F, conv.kernel)


def _fold(bn, device):
    """Eval-mode BatchNorm as (scale, shift) f32 [C]: y = x * scale + shift.  Cached on the module,
    keyed by the version counters of its parameters and running statistics (they do not change
    between inference calls)."""
    # (running statistics are written by the training kernels in place: the epoch covers them too)
    key = (B.weights_key(bn.weight), B.weights_key(bn.bias), B.weights_key(bn.running_mean),
           B.weights_key(bn.running_var), str(device))
    cached = getattr(bn, '_lidal_fold', None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    c = bn.num_features
    fold = torch.empty((2, c), dtype=torch.float32, device=device)
    B.check(B.lib().lidal_bn_fold(B.ptr(bn.weight), B.ptr(bn.bias), B.ptr(bn.running_mean),
                                  B.ptr(bn.running_var), float(bn.eps), c, B.ptr(fold[0]),
                                  B.ptr(fold[1]), B.stream()), 'bn_fold')
    object.__setattr__(bn, '_lidal_fold', (key, fold[0], fold[1]))
    return fold[0], fold[1]


def _conv_norm(conv, bn, x, residual=None, relu_after=False):
    scale, shift = _fold(bn, x.F.device)
    relu = int(bn.fused_relu) | (2 if (relu_after and residual is not None) else 0)
    return conv3d(x, conv.kernel, kernel_size=conv.kernel_size, bias=None, stride=conv.stride,
                  dilation=conv.dilation, transposed=conv.transposed,
                  epilogue=(scale, shift, relu, residual))


def _conv_bn(inc, outc, ks, stride=1, transposed=False):
    conv = spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, dilation=1, transposed=transposed)
    conv.bn_follows = True      # training: the conv kernel leaves the BatchNorm's batch statistics on its output
    return [conv, spnn.BatchNorm(outc)]


def conv_bn_relu(inc, outc, ks, stride=1, transposed=False):
    """Conv3d -> BatchNorm -> ReLU as three Sequential slots (indices 0, 1, 2 as in the reference, so
    `net.1.weight` etc. keep their names); the ReLU is computed inside the BatchNorm kernels and
    slot 2 is an Identity."""
    conv, bn = _conv_bn(inc, outc, ks, stride, transposed)
    bn.fused_relu = True
    return [conv, bn, nn.Identity()]


class BasicConvolutionBlock(nn.Module):
    """Conv3d -> BatchNorm -> ReLU."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = ConvNormSequential(*conv_bn_relu(inc, outc, ks, stride))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):
    """transposed Conv3d -> BatchNorm -> ReLU."""

    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = ConvNormSequential(*conv_bn_relu(inc, outc, ks, stride, transposed=True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    """relu(conv-bn-relu-conv-bn(x) + shortcut(x)); 1x1 conv + BN shortcut iff the shape changes."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = ConvNormSequential(*conv_bn_relu(inc, outc, ks, stride), *_conv_bn(outc, outc, ks, 1))
        if inc == outc and stride == 1:
            self.downsample = nn.Identity()
        else:
            self.downsample = ConvNormSequential(*_conv_bn(inc, outc, 1, stride))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        first = self.net[0]
        if ((B.FORK & 1) and B.wants_grad(x.F) and x.F.requires_grad and first.bias is None
                and first.kernel_size != (1, 1, 1) and not first.transposed):
            # training: x feeds the first convolution AND the shortcut; fork it inside that
            # convolution so the shortcut's gradient is added in the data-gradient kernel's epilogue
            # (16 blocks: 16 fewer passes of autograd's gradient accumulation per step)
            y, x_skip = self.net[0](x, fork=True)
            b = self.downsample(x_skip)
            return self.net(y, residual=b.F, relu_after=True, start=1)
        b = self.downsample(x)
        # == self.relu(self.net(x) + b): one add+ReLU pass in training, none at inference (the sum
        # and the ReLU run in the epilogue of net's last convolution)
        return self.net(x, residual=b.F, relu_after=True)
