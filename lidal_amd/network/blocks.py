"""Building blocks of the sparse U-Net (counterparts of /root/reference/network/utils.py:105-172).

Attribute names (`net`, `downsample`, `relu`) and Sequential positions are the checkpoint
surface: state_dict keys such as `stage2.1.net.3.kernel` or `up1.1.0.downsample.1.weight` must
match the reference's so its checkpoints load with strict=True (train.py:66, prob_inference.py:68).
"""
from torch import nn

from .. import SparseTensor
from .. import nn as spnn
from ..nn.functional.fused import add_relu


def _conv_bn(inc, outc, ks, stride=1, transposed=False):
    return [spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, dilation=1,
                        transposed=transposed),
            spnn.BatchNorm(outc)]


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
        self.net = nn.Sequential(*conv_bn_relu(inc, outc, ks, stride))

    def forward(self, x):
        return self.net(x)


class BasicDeconvolutionBlock(nn.Module):
    """transposed Conv3d -> BatchNorm -> ReLU."""

    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(*conv_bn_relu(inc, outc, ks, stride, transposed=True))

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    """relu(conv-bn-relu-conv-bn(x) + shortcut(x)); 1x1 conv + BN shortcut iff the shape changes."""

    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(*conv_bn_relu(inc, outc, ks, stride), *_conv_bn(outc, outc, ks, 1))
        if inc == outc and stride == 1:
            self.downsample = nn.Identity()
        else:
            self.downsample = nn.Sequential(*_conv_bn(inc, outc, 1, stride))
        self.relu = spnn.ReLU(True)

    def forward(self, x):
        a, b = self.net(x), self.downsample(x)
        out = SparseTensor(add_relu(a.F, b.F), a.C, a.s)       # == self.relu(a + b), one pass
        out.cmaps, out.kmaps = a.cmaps, a.kmaps
        return out
