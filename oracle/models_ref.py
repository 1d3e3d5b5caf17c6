"""SPVCNN / MinkUNet restated on the CPU oracle operators.  TEST INFRASTRUCTURE.

/root/reference/network/{spvcnn,minkunet,utils}.py cannot travel to the GPU box, so model-level
parity there (and bench.py's cpu_baseline) needs the architecture on oracle.tsref.  This file
follows network/spvcnn.py:11-155, network/minkunet.py:15-122 and network/utils.py:13-172 and is
PINNED in the build container: tests/golden/make_golden.py asserts that it reproduces the
unchanged reference files' logits bit for bit on the same weights and input.

The architecture is written against the torchsparse SURFACE only (SparseTensor / PointTensor / cat,
nn.Conv3d / BatchNorm / ReLU composed with nn.Sequential, nn.functional.sp*, get_kernel_offsets), as
the reference's files are; `build_models(package)` instantiates it over any package exposing that
surface: `oracle.tsref` gives the CPU oracle (MinkUNetRef / SPVCNNRef below); the GPU tests also
instantiate it over `lidal_amd` to run a U-Net that uses none of lidal_amd.network's fused paths.
"""
import torch
from torch import nn

from oracle import tsref

CS = [32, 32, 64, 128, 256, 256, 128, 96, 96]


# ---- bf16 emulation (round 4) -------------------------------------------------------------------------------
# The benchmarked dtype is "bf16 operands, f32 accumulation": the HIP path STORES every activation and every
# activation gradient in bf16 and computes between the storage points in f32 / f64.  Its fused kernels round exactly
# as the separate operators would (a sum fused into a BatchNorm or into a convolution epilogue first rounds the
# summand it produced, csrc/bn.hip bn_apply_kernel, csrc/conv_img.hip store_tile, csrc/voxel.hip), which is why they
# are bitwise the per-operator path -- so the emulation is simply: ROUND THE OUTPUT OF EVERY OPERATOR, forward and
# backward.  `emulate_bf16(model)` makes an oracle model (run it in float64) do that:
#   * every Conv3d / Linear / BatchNorm output, every voxelize / devoxelize output, every sum (residual block,
#     point branch), the logits: rounded; in the backward pass the gradient arriving at that tensor is rounded (the
#     sum over its consumers, as the HIP path stores it);
#   * the gradient each Conv3d / Linear / BatchNorm / voxelize / devoxelize hands to its input is rounded when it
#     leaves the operator (before autograd adds another consumer's to it);
#   * weights of Conv3d / Linear: rounded on the way into the product (the LDS images are bf16), their gradient
#     passed through unrounded (accumulated and kept in f32);
#   * BatchNorm statistics, the loss, the parameter gradients: not rounded (f32 / f64 in the HIP path).
# All helpers are the identity unless emulation was switched on for the model instance, so the pinned f32 / f64
# behaviour of this file (tests/golden/make_golden.py) is untouched.
class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


_EMULATE = [False]          # set around the forward pass of a model emulate_bf16() was applied to


def _rb(x):
    return _Round.apply(x) if _EMULATE[0] else x


def _rb_bwd(x):
    return _RoundBwd.apply(x) if (_EMULATE[0] and x.requires_grad) else x


def _feats(t, fn):
    """fn applied to the features of a SparseTensor-like `t` (same coordinates and caches), or to a plain tensor."""
    if torch.is_tensor(t):
        return fn(t)
    out = type(t)(fn(t.feats), t.coords, t.stride)
    out.cmaps, out.kmaps = t.cmaps, t.kmaps
    return out


def emulate_bf16(model):
    """Switch the bf16 storage emulation on for this model INSTANCE (see above).  Returns the model."""
    conv_t = type(model.stem[0])
    bn_t = type(model.stem[1])
    res_t = type(model.stage1[1])

    def edge(mod, args):                                # the gradient this operator hands to its input: rounded
        return (_feats(args[0], _rb_bwd),) + tuple(args[1:])

    def stored(mod, args, out):                         # the operator's output: rounded (and the gradient arriving at it)
        return _feats(out, _Round.apply)
    for m in model.modules():
        if isinstance(m, (conv_t, nn.Linear)):
            name = 'kernel' if isinstance(m, conv_t) else 'weight'

            def pre(mod, args, name=name):              # the operand of the product is the bf16-rounded weight
                w = mod._parameters[name]
                mod._emul_saved = w
                del mod._parameters[name]
                setattr(mod, name, _RoundFwd.apply(w))
                return edge(mod, args)

            def post(mod, args, out, name=name):
                delattr(mod, name)
                mod._parameters[name] = mod._emul_saved
                del mod._emul_saved
                return _feats(out, _Round.apply)
            m.register_forward_pre_hook(pre)
            m.register_forward_hook(post)
        elif isinstance(m, (bn_t, nn.BatchNorm1d)):
            m.register_forward_pre_hook(edge)
            m.register_forward_hook(stored)
        elif isinstance(m, res_t):                      # relu(net(x) + shortcut(x)): the sum is an operator output
            m.register_forward_hook(stored)

    def enter(mod, args):
        _EMULATE[0] = True
        x = args[0]
        return (_feats(x, lambda f: f.detach().to(torch.bfloat16).to(f.dtype)),)        # the stem's operand is bf16

    def leave(mod, args, out):
        _EMULATE[0] = False
    model.register_forward_pre_hook(enter)
    model.register_forward_hook(leave, always_call=True)
    return model


def build_models(ts):
    """-> (MinkUNet class, SPVCNN class) over the torchsparse-like package `ts`."""
    import importlib
    spnn = importlib.import_module(ts.__name__ + '.nn')
    F = importlib.import_module(ts.__name__ + '.nn.functional')
    get_kernel_offsets = importlib.import_module(ts.__name__ + '.nn.utils').get_kernel_offsets


    def _cb(i, o, ks, s=1, t=False):
        return [spnn.Conv3d(i, o, kernel_size=ks, stride=s, transposed=t), spnn.BatchNorm(o)]


    class _Seq(nn.Module):                      # network/utils.py:105-139 (conv / deconv blocks)
        def __init__(self, layers):
            super().__init__()
            self.net = nn.Sequential(*layers)

        def forward(self, x):
            return self.net(x)


    class _Res(nn.Module):                      # network/utils.py:142-172
        def __init__(self, i, o):
            super().__init__()
            self.net = nn.Sequential(*_cb(i, o, 3), spnn.ReLU(True), *_cb(o, o, 3))
            self.downsample = nn.Identity() if i == o else nn.Sequential(*_cb(i, o, 1))
            self.relu = spnn.ReLU(True)

        def forward(self, x):
            return self.relu(self.net(x) + self.downsample(x))


    class _UNet(nn.Module):
        def __init__(self, class_num):
            super().__init__()
            cs = CS
            self.stem = nn.Sequential(*_cb(4, cs[0], 3), spnn.ReLU(True), *_cb(cs[0], cs[0], 3),
                                      spnn.ReLU(True))
            for i in range(1, 5):
                setattr(self, 'stage%d' % i, nn.Sequential(
                    _Seq(_cb(cs[i - 1], cs[i - 1], 2, 2) + [spnn.ReLU(True)]),
                    _Res(cs[i - 1], cs[i]), _Res(cs[i], cs[i])))
            for i in range(1, 5):
                setattr(self, 'up%d' % i, nn.ModuleList([
                    _Seq(_cb(cs[3 + i], cs[4 + i], 2, 2, True) + [spnn.ReLU(True)]),
                    nn.Sequential(_Res(cs[4 + i] + cs[4 - i], cs[4 + i]), _Res(cs[4 + i], cs[4 + i]))]))
            self.classifier = nn.Sequential(nn.Linear(cs[8], class_num))

        @staticmethod
        def _up(stage, y, skip):
            if _EMULATE[0]:         # bf16 emulation: the gradient of the concatenation (the sum over its consumers) is stored
                return stage[1](_feats(ts.cat([stage[0](y), skip]), _rb_bwd))
            return stage[1](ts.cat([stage[0](y), skip]))


    class MinkUNetRef(_UNet):                   # network/minkunet.py:97-122
        def forward(self, x):
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


    def _floor_to_stride(z, s):
        return torch.cat([torch.floor(z.C[:, :3] / s).int() * s, z.C[:, -1].int().view(-1, 1)], 1)


    def initial_voxelize(z, init_res, after_res):        # network/utils.py:13-33
        nc = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
        pc_hash = F.sphash(torch.floor(nc).int())
        sparse_hash = torch.unique(pc_hash)
        idx_query = F.sphashquery(pc_hash, sparse_hash)
        counts = F.spcount(idx_query.int(), len(sparse_hash))
        coords = torch.round(F.spvoxelize(torch.floor(nc), idx_query, counts)).int()
        x = ts.SparseTensor(F.spvoxelize(z.F, idx_query, counts), coords, 1)
        x.cmaps.setdefault(x.stride, x.coords)
        z.C = nc
        return x


    def point_to_voxel(x, z):                            # network/utils.py:38-61
        ci, cc = z.additional_features['idx_query'], z.additional_features['counts']
        if ci.get(x.s) is None:
            idx_query = F.sphashquery(F.sphash(_floor_to_stride(z, x.s[0])), F.sphash(x.C))
            ci[x.s] = idx_query
            cc[x.s] = F.spcount(idx_query.int(), x.C.shape[0])
        out = ts.SparseTensor(_rb(F.spvoxelize(_rb_bwd(z.F), ci[x.s], cc[x.s])), x.C, x.s)      # (_rb*: bf16 emulation, else identity)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out


    def voxel_to_point(x, z):                            # network/utils.py:66-102
        if z.idx_query.get(x.s) is None:
            off = get_kernel_offsets(2, x.s, 1, device=z.F.device)
            idx_query = F.sphashquery(F.sphash(_floor_to_stride(z, x.s[0]), off), F.sphash(x.C))
            z.weights[x.s] = F.calc_ti_weights(z.C, idx_query, scale=x.s[0]).transpose(0, 1).contiguous()
            z.idx_query[x.s] = idx_query.transpose(0, 1).contiguous()
        out = ts.PointTensor(_rb(F.spdevoxelize(_rb_bwd(x.F), z.idx_query[x.s], z.weights[x.s])), z.C,
                                idx_query=z.idx_query, weights=z.weights)
        out.additional_features = z.additional_features
        return out


    class SPVCNNRef(_UNet):                     # network/spvcnn.py:11-155
        def __init__(self, class_num):
            super().__init__(class_num)
            cs = CS
            self.point_transforms = nn.ModuleList([
                nn.Sequential(nn.Linear(a, b), nn.BatchNorm1d(b), nn.ReLU(True))
                for a, b in ((cs[0], cs[4]), (cs[4], cs[6]), (cs[6], cs[8]))])
            self.dropout = nn.Dropout(0.3, True)

        def forward(self, x):
            z = ts.PointTensor(x.F, x.C.float())
            x0 = self.stem(initial_voxelize(z, 0.05, 0.05))
            z0 = voxel_to_point(x0, z)
            x1 = self.stage1(point_to_voxel(x0, z0))
            x2 = self.stage2(x1)
            x3 = self.stage3(x2)
            x4 = self.stage4(x3)
            z1 = voxel_to_point(x4, z0)
            z1.F = _rb(z1.F + self.point_transforms[0](z0.F))       # (_rb: bf16 emulation, else identity)
            y1 = point_to_voxel(x4, z1)
            y1.F = self.dropout(y1.F)
            y1 = self._up(self.up1, y1, x3)
            y2 = self._up(self.up2, y1, x2)
            z2 = voxel_to_point(y2, z1)
            z2.F = _rb(z2.F + self.point_transforms[1](z1.F))
            y3 = point_to_voxel(y2, z2)
            y3.F = self.dropout(y3.F)
            y3 = self._up(self.up3, y3, x1)
            y4 = self._up(self.up4, y3, x0)
            z3 = voxel_to_point(y4, z2)
            z3.F = _rb(z3.F + self.point_transforms[2](z2.F))
            return self.classifier(z3.F), z3.F

    return MinkUNetRef, SPVCNNRef


MinkUNetRef, SPVCNNRef = build_models(tsref)
