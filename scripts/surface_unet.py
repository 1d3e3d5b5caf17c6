"""A LiDAL user's network as it reaches this package through the drop-in boundary: SPVCNN / MinkUNet composed ONLY
of the torchsparse-named surface (`SparseTensor`, `PointTensor`, `cat`, `nn.Conv3d`, `nn.BatchNorm`, `nn.ReLU(True)`,
`nn.functional.sp*`, `nn.utils.get_kernel_offsets`) inside plain `torch.nn.Sequential`s -- the composition of
/root/reference/network/utils.py:105-172 and network/spvcnn.py:112-155, with the reference's state_dict keys -- and none of
`lidal_amd.network` (no launch plan, no fused block, no prefetched geometry, no flag set on any module).

`bench.py` times it as `variants.dropin_surface` (what `install_as_torchsparse()` users get), and
tests/test_model_gpu.py checks it against the package's own networks.  `build(ts)` takes the package that plays
torchsparse (lidal_amd, or anything with the same surface)."""
import importlib

import torch
from torch import nn

WIDTHS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


def build(ts):
    """-> {'minkunet': class, 'spvcnn': class} over the torchsparse-like package `ts`."""
    spnn = importlib.import_module(ts.__name__ + '.nn')
    F = importlib.import_module(ts.__name__ + '.nn.functional')
    kernel_offsets = importlib.import_module(ts.__name__ + '.nn.utils').get_kernel_offsets

    def unit(cin, cout, ks, stride=1, transposed=False, act=True):
        mods = [spnn.Conv3d(cin, cout, kernel_size=ks, stride=stride, transposed=transposed), spnn.BatchNorm(cout)]
        return mods + [spnn.ReLU(True)] if act else mods

    class Wrapped(nn.Module):               # the reference keeps conv / deconv blocks under `.net`
        def __init__(self, mods):
            super().__init__()
            self.net = nn.Sequential(*mods)

        def forward(self, x):
            return self.net(x)

    class Residual(nn.Module):
        def __init__(self, cin, cout):
            super().__init__()
            self.net = nn.Sequential(*unit(cin, cout, 3), *unit(cout, cout, 3, act=False))
            self.downsample = nn.Sequential(*unit(cin, cout, 1, act=False)) if cin != cout else nn.Identity()
            self.relu = spnn.ReLU(True)

        def forward(self, x):
            return self.relu(self.net(x) + self.downsample(x))

    class UNet(nn.Module):
        def __init__(self, class_num):
            super().__init__()
            w = WIDTHS
            self.stem = nn.Sequential(*unit(4, w[0], 3), *unit(w[0], w[0], 3))
            for i in range(4):
                self.add_module('stage%d' % (i + 1), nn.Sequential(
                    Wrapped(unit(w[i], w[i], 2, stride=2)), Residual(w[i], w[i + 1]), Residual(w[i + 1], w[i + 1])))
            for i in range(4):
                self.add_module('up%d' % (i + 1), nn.ModuleList([
                    Wrapped(unit(w[4 + i], w[5 + i], 2, stride=2, transposed=True)),
                    nn.Sequential(Residual(w[5 + i] + w[3 - i], w[5 + i]), Residual(w[5 + i], w[5 + i]))]))
            self.classifier = nn.Sequential(nn.Linear(w[8], class_num))

        def encode(self, x0):
            xs = [x0]
            for i in range(4):
                xs.append(getattr(self, 'stage%d' % (i + 1))(xs[-1]))
            return xs

        def decode_step(self, i, y, skip):
            up = getattr(self, 'up%d' % i)
            return up[1](ts.cat([up[0](y), skip]))

    class MinkUNet(UNet):
        def forward(self, x):
            xs = self.encode(self.stem(x))
            y = xs[4]
            for i in range(1, 5):
                y = self.decode_step(i, y, xs[4 - i])
            return self.classifier(y.F), y.F

    # ---- point <-> voxel glue (network/utils.py:13-102), on the surface's functional operators
    def snap(z, s):
        return torch.cat([torch.floor(z.C[:, :3] / s).int() * s, z.C[:, -1].int().view(-1, 1)], 1)

    def first_voxels(z, init_res, after_res):
        pc = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
        h = F.sphash(torch.floor(pc).int())
        uniq = torch.unique(h)
        q = F.sphashquery(h, uniq)
        cnt = F.spcount(q.int(), len(uniq))
        coords = torch.round(F.spvoxelize(torch.floor(pc), q, cnt)).int()
        x = ts.SparseTensor(F.spvoxelize(z.F, q, cnt), coords, 1)
        x.cmaps.setdefault(x.stride, x.coords)
        z.additional_features['idx_query'][1] = q
        z.additional_features['counts'][1] = cnt
        z.C = pc
        return x

    def to_voxels(x, z):
        qs, cs = z.additional_features['idx_query'], z.additional_features['counts']
        if qs.get(x.s) is None:
            q = F.sphashquery(F.sphash(snap(z, x.s[0])), F.sphash(x.C))
            qs[x.s], cs[x.s] = q, F.spcount(q.int(), x.C.shape[0])
        out = ts.SparseTensor(F.spvoxelize(z.F, qs[x.s], cs[x.s]), x.C, x.s)
        out.cmaps, out.kmaps = x.cmaps, x.kmaps
        return out

    def to_points(x, z):
        if z.idx_query.get(x.s) is None:
            off = kernel_offsets(2, x.s, 1, device=z.F.device)
            q = F.sphashquery(F.sphash(snap(z, x.s[0]), off), F.sphash(x.C))
            z.weights[x.s] = F.calc_ti_weights(z.C, q, scale=x.s[0]).transpose(0, 1).contiguous()
            z.idx_query[x.s] = q.transpose(0, 1).contiguous()
        out = ts.PointTensor(F.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s]), z.C, idx_query=z.idx_query,
                             weights=z.weights)
        out.additional_features = z.additional_features
        return out

    class SPVCNN(UNet):
        def __init__(self, class_num):
            super().__init__(class_num)
            w = WIDTHS
            self.point_transforms = nn.ModuleList([nn.Sequential(nn.Linear(a, b), nn.BatchNorm1d(b), nn.ReLU(True))
                                                   for a, b in ((w[0], w[4]), (w[4], w[6]), (w[6], w[8]))])
            self.dropout = nn.Dropout(0.3, True)

        def forward(self, x):
            z = ts.PointTensor(x.F, x.C.float())
            x0 = self.stem(first_voxels(z, 0.05, 0.05))
            z0 = to_points(x0, z)
            xs = self.encode(to_voxels(x0, z0))
            xs[0] = x0
            z1 = to_points(xs[4], z0)
            z1.F = z1.F + self.point_transforms[0](z0.F)
            y = to_voxels(xs[4], z1)
            y.F = self.dropout(y.F)
            y = self.decode_step(2, self.decode_step(1, y, xs[3]), xs[2])
            z2 = to_points(y, z1)
            z2.F = z2.F + self.point_transforms[1](z1.F)
            y = to_voxels(y, z2)
            y.F = self.dropout(y.F)
            y = self.decode_step(4, self.decode_step(3, y, xs[1]), xs[0])
            z3 = to_points(y, z2)
            z3.F = z3.F + self.point_transforms[2](z2.F)
            return self.classifier(z3.F), z3.F

    return {'minkunet': MinkUNet, 'spvcnn': SPVCNN}
