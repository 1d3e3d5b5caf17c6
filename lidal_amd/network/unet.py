"""SPVCNN and MinkUNet on the HIP operators (counterparts of /root/reference/network/spvcnn.py
and network/minkunet.py).

Both models are the same 4-down / 4-up sparse U-Net (channels 32-32-64-128-256-256-128-96-96,
spvcnn.py:15); SPVCNN adds a point branch (three Linear+BN+ReLU "point transforms") that
exchanges features with the voxel branch at four places.  Module names and Sequential indices
reproduce the reference's state_dict keys (tests/golden/state_dict_*.json).
"""
from torch import nn

from .. import PointTensor, cat
from .. import nn as spnn
from .blocks import (BasicConvolutionBlock, BasicDeconvolutionBlock, ConvNormSequential,
                     ResidualBlock, conv_bn_relu)
from ..nn.functional.conv import prefetch_kernel_maps
from .glue import initial_voxelize, point_to_voxel, voxel_to_point
from .plan import planned_forward

__all__ = ['SPVCNN', 'MinkUNet']

CHANNELS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


class _SparseUNet(nn.Module):
    """stem, stage1-4, up1-4, classifier -- shared by both backbones."""

    # (kernel_size, stride) of the encoder's convs in data-flow order: stem 3x3x3, then per stage
    # a 2x2x2 stride-2 conv and 3x3x3 residual blocks.  Used to build every kernel map up front.
    MAP_PLAN = ((3, 1),) + ((2, 2), (3, 1)) * 4

    def __init__(self, class_num, cr=1.0):
        super().__init__()
        cs = [int(cr * c) for c in CHANNELS]
        self.cs = cs
        self.num_classes = class_num        # (the scorer sizes its exchange buffers from it before any inference)
        self.stem = ConvNormSequential(*conv_bn_relu(4, cs[0], 3), *conv_bn_relu(cs[0], cs[0], 3))
        for i in range(1, 5):          # encoder: stride-2 conv then two residual blocks
            setattr(self, 'stage%d' % i, nn.Sequential(
                BasicConvolutionBlock(cs[i - 1], cs[i - 1], ks=2, stride=2, dilation=1),
                ResidualBlock(cs[i - 1], cs[i], ks=3, stride=1, dilation=1),
                ResidualBlock(cs[i], cs[i], ks=3, stride=1, dilation=1)))
        for i in range(1, 5):          # decoder: transposed conv, concat skip, two residual blocks
            skip = cs[4 - i]
            setattr(self, 'up%d' % i, nn.ModuleList([
                BasicDeconvolutionBlock(cs[3 + i], cs[4 + i], ks=2, stride=2),
                nn.Sequential(ResidualBlock(cs[4 + i] + skip, cs[4 + i], ks=3, stride=1, dilation=1),
                              ResidualBlock(cs[4 + i], cs[4 + i], ks=3, stride=1, dilation=1))]))
        self.classifier = nn.Sequential(spnn.Linear(cs[8], class_num))

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _up(stage, y, skip):
        y = stage[0](y)
        return stage[1](cat([y, skip]))


class MinkUNet(_SparseUNet):
    def __init__(self, class_num):
        super().__init__(class_num)
        self.weight_initialization()

    def forward(self, x):
        out = planned_forward(self, x)              # training: the whole pass as one launch plan (network/plan.py)
        if out is not None:
            return out
        g = getattr(x, 'geometry', None)            # tables built ahead of the features (network/geometry.py)
        if g is not None:
            x, _ = g.enter(x, 'MinkUNet')
        else:
            prefetch_kernel_maps(x, self.MAP_PLAN)
        x0 = self.stem(x)
        x1 = self.stage1(x0)
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        y = self._up(self.up1, x4, x3)
        y = self._up(self.up2, y, x2)
        y = self._up(self.up3, y, x1)
        y = self._up(self.up4, y, x0)
        return self.classifier(y.F), y.F


class SPVCNN(_SparseUNet):
    def __init__(self, class_num):
        super().__init__(class_num)
        cs = self.cs
        self.pres = 0.05
        self.vres = 0.05
        self.point_transforms = nn.ModuleList([
            ConvNormSequential(spnn.Linear(a, b), spnn.BatchNorm1d(b), nn.Identity())   # ReLU fused in BN
            for a, b in ((cs[0], cs[4]), (cs[4], cs[6]), (cs[6], cs[8]))])
        for seq in self.point_transforms:
            seq[0].bn_follows = True        # the Linear kernel leaves the BatchNorm's batch statistics
            seq[1].fused_relu = True
        self.weight_initialization()
        self.dropout = nn.Dropout(0.3, True)

    # strides at which features cross between the point and the voxel branch (forward below)
    POINT_STRIDES = (1, 16, 4)

    def forward(self, x):
        out = planned_forward(self, x)              # training: the whole pass as one launch plan (network/plan.py)
        if out is not None:
            return out
        g = getattr(x, 'geometry', None)            # tables built ahead of the features (network/geometry.py)
        if g is not None:
            x0, z = g.enter(x, 'SPVCNN')
            x0 = self.stem(x0)
        else:
            z = PointTensor(x.F, x.C.float())
            x0 = self.stem(prefetch_kernel_maps(initial_voxelize(z, self.pres, self.vres), self.MAP_PLAN))
        z0 = voxel_to_point(x0, z, nearest=False, own_cells=True)       # (x0 and its pyramid derive from z: glue.corner_tables)

        x1 = self.stage1(point_to_voxel(x0, z0))
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        z1 = voxel_to_point(x4, z0, own_cells=True)
        z1.F = self.point_transforms[0](z0.F, residual=z1.F)          # z1.F + transform(z0.F)

        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self._up(self.up1, y1, x3)
        y2 = self._up(self.up2, y1, x2)
        z2 = voxel_to_point(y2, z1, own_cells=True)
        z2.F = self.point_transforms[1](z1.F, residual=z2.F)

        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self._up(self.up3, y3, x1)
        y4 = self._up(self.up4, y3, x0)
        z3 = voxel_to_point(y4, z2, own_cells=True)
        z3.F = self.point_transforms[2](z2.F, residual=z3.F)
        return self.classifier(z3.F), z3.F
