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
from ..utils import make_ntuple


# ---- training: whole blocks as ONE autograd node ----------------------------------------------
# A train step of one scan is host-bound (12.8 ms wall for ~9 ms of kernels; scripts/profile_host.py:
# forward 7.5 ms of Python / autograd / launch calls against 3.5 ms of GPU work).  Every Conv3d,
# BatchNorm, add+ReLU used to be an autograd Function of its own -- ~100 us of Python and engine
# time per Conv3d -> BatchNorm pair.  The Functions below run a whole Conv3d -> BatchNorm -> ReLU unit
# resp. a whole residual block (network/utils.py:105-172) with the SAME raw operator calls
# (nn/functional/{conv,norm,dense}.py: _forward / conv_backward / train_forward / train_backward /
# rows_backward) in the same order: results are bitwise those of the per-operator path
# (tests/test_model_gpu.py::test_fused_block_functions_are_bitwise_the_per_operator_path), with one
# node instead of three resp. up to twelve.  LIDAL_FUSE_BLOCKS=0 keeps the per-operator path.
import os as _os

from ..nn.functional import conv as _C
from ..nn.functional import dense as _D
from ..nn.functional import norm as _N

FUSE_BLOCKS = _os.environ.get('LIDAL_FUSE_BLOCKS', '1') != '0'
BN_SUMS = _os.environ.get('LIDAL_BN_SUMS', '1') != '0'       # BatchNorm backward sums from the data-gradient launches


def _bn_args(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)


def _bn_ok(bn, feats):
    return (bn.training and bn.track_running_stats and bn.momentum is not None and bn.weight is not None
            and bn.bias is not None and bn.weight.dtype == torch.float32 and feats.is_cuda)


class _ConvNormAct(torch.autograd.Function):
    """Conv3d (regular / strided / transposed) -> train-mode BatchNorm -> ReLU."""

    @staticmethod
    def forward(ctx, feats, kernel, gamma, beta, kmap, transposed, bn, relu, stats):
        need_gx = ctx.needs_input_grad[0]
        xc, x1, ctx.img_bwd = _C._forward(feats, kernel, kmap, transposed, None, need_gx, stats)
        y, mean, invstd, x1, w, b = _N.train_forward(x1, gamma, beta, bn.running_mean, bn.running_var, bn.momentum,
                                                     bn.eps, relu, bn.num_batches_tracked,
                                                     getattr(x1, '_lidal_bn_stats', None))
        ctx.kmap, ctx.transposed, ctx.relu = kmap, transposed, relu
        ctx.save_for_backward(xc, kernel, x1, w, b, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, g):
        B.note_backward()
        xc, kernel, x1, w, b, mean, invstd = ctx.saved_tensors
        dx, gg, gb, _ = _N.train_backward(x1, w, b, mean, invstd, ctx.relu, g, True)
        gx, gk = _C.conv_backward(xc, kernel, ctx.kmap, ctx.transposed, ctx.img_bwd, dx, None,
                                  ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gk, gg, gb, None, None, None, None, None


class _Residual(torch.autograd.Function):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut(x)), shortcut = identity or bn_s(x @ ks)."""

    @staticmethod
    def forward(ctx, feats, k1, g1, b1, k2, g2, b2, ks, gs, bs, kmap, bn1, bn2, bns, stats):
        # `stats` = (s1, s2, ss): does the kernel of conv1 / conv2 / the shortcut leave its BatchNorm's statistics?
        need_gx = ctx.needs_input_grad[0]
        xc, x1, ctx.img1 = _C._forward(feats, k1, kmap, False, None, need_gx, stats[0])
        y1, mean1, inv1, x1, w1, c1 = _N.train_forward(x1, g1, b1, bn1.running_mean, bn1.running_var, bn1.momentum,
                                                       bn1.eps, True, bn1.num_batches_tracked,
                                                       getattr(x1, '_lidal_bn_stats', None))
        _, x2, ctx.img2 = _C._forward(y1, k2, kmap, False, None, True, stats[1])
        ctx.shortcut = ks is not None
        if ctx.shortcut:
            xs_in, ctx.wcs, ctx.pads, xs, ctx.imgs = _D._forward(feats, ks, None, False, None, need_gx, stats[2])
            res, means, invs, xs, ws, cs = _N.train_forward(xs, gs, bs, bns.running_mean, bns.running_var,
                                                            bns.momentum, bns.eps, False, bns.num_batches_tracked,
                                                            getattr(xs, '_lidal_bn_stats', None))
        else:
            res = feats
        out, mean2, inv2, x2, w2, c2 = _N.train_forward(x2, g2, b2, bn2.running_mean, bn2.running_var, bn2.momentum,
                                                        bn2.eps, False, bn2.num_batches_tracked,
                                                        getattr(x2, '_lidal_bn_stats', None), res, True)
        ctx.kmap = kmap
        saved = [xc, k1, x1, w1, c1, mean1, inv1, y1, k2, x2, w2, c2, mean2, inv2, out]
        if ctx.shortcut:
            saved += [xs_in, ks, xs, ws, cs, means, invs]
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, g):
        B.note_backward()
        t = ctx.saved_tensors
        xc, k1, x1, w1, c1, mean1, inv1, y1, k2, x2, w2, c2, mean2, inv2, out = t[:15]
        need_gx = ctx.needs_input_grad[0]
        # relu(bn2 + shortcut): the gradient where the output is positive, for both summands
        gks = ggs = gbs = None
        if ctx.shortcut:
            xs_in, ks, xs, ws, cs, means, invs = t[15:]
            gm, (dx2, gg2, gb2), (dxs, ggs, gbs) = _N.tail_backward(g, out, x2, w2, c2, mean2, inv2, (xs, ws, cs, means, invs))
            g_skip, gks, _ = _D.rows_backward(xs_in, ks, ctx.wcs, ctx.imgs, ctx.pads, False, dxs, need_gx,
                                              ctx.needs_input_grad[7], False)
        else:
            gm, (dx2, gg2, gb2), _ = _N.tail_backward(g, out, x2, w2, c2, mean2, inv2)
            g_skip = gm
        # conv2's data gradient IS the output gradient of bn1 (y1 has no other consumer): the launch that writes it
        # also leaves bn1's backward sums per tile (BN_SUMS; bf16), and bn1's backward skips its pass for them
        dy1, gk2 = _C.conv_backward(y1, k2, ctx.kmap, False, ctx.img2, dx2, None, True, ctx.needs_input_grad[4],
                                    (x1, mean1, inv1, w1, c1, True) if BN_SUMS else None)
        dx1, gg1, gb1, _ = _N.train_backward(x1, w1, c1, mean1, inv1, True, dy1, True, None,
                                             getattr(dy1, '_lidal_bnb_sums', None))
        gx, gk1 = _C.conv_backward(xc, k1, ctx.kmap, False, ctx.img1, dx1, g_skip if need_gx else None, need_gx,
                                   ctx.needs_input_grad[1])
        return gx, gk1, gg1, gb1, gk2, gg2, gb2, gks, ggs, gbs, None, None, None, None, None


def _kmap_of(x, conv):
    """(kernel map, output coords, output stride) of a non-1x1x1 Conv3d on x: looked up in x.kmaps, built on a miss
    (as F.conv3d does)."""
    from ..nn.functional.conv import build_kernel_map
    if conv.transposed:
        ts = tuple(x.stride[k] // conv.stride[k] for k in range(3))
        return x.kmaps[(ts, conv.kernel_size, conv.stride, (1, 1, 1))], x.cmaps[ts], ts
    key = (x.stride, conv.kernel_size, conv.stride, (1, 1, 1))
    out_stride = tuple(x.stride[k] * conv.stride[k] for k in range(3))
    kmap = x.kmaps.get(key)
    if kmap is None:
        kmap, out_coords = build_kernel_map(x.coords, x.stride, conv.kernel_size, conv.stride, x.cmaps)
        x.kmaps[key] = kmap
        if any(s > 1 for s in conv.stride):
            x.cmaps.setdefault(out_stride, out_coords)
    out_coords = x.coords if all(s == 1 for s in conv.stride) else x.cmaps[out_stride]
    return kmap, out_coords, out_stride


def _wrap(feats, coords, stride, like):
    out = SparseTensor(feats, coords, stride)
    out.cmaps, out.kmaps = like.cmaps, like.kmaps
    out.cmaps.setdefault(out.stride, out.coords)
    return out


def _plain_conv(conv):
    d = conv.dilation
    return (conv.bias is None and conv.kernel_size != (1, 1, 1) and (d == 1 or tuple(make_ntuple(d, 3)) == (1, 1, 1)))


def fused_conv_norm_act(x, conv, bn):
    """Conv3d -> BatchNorm -> ReLU as one autograd node (training), or None if the configuration is not
    the standard one (then the caller runs the modules one by one)."""
    if not (FUSE_BLOCKS and _plain_conv(conv) and _bn_ok(bn, x.F) and B.wants_grad(x.F, conv.kernel, bn.weight)
            and isinstance(bn, spnn.BatchNorm)):
        return None
    kmap, coords, stride = _kmap_of(x, conv)
    y = _ConvNormAct.apply(x.F, conv.kernel, bn.weight, bn.bias, kmap, conv.transposed, bn, bool(bn.fused_relu),
                           bool(conv.bn_follows))
    return _wrap(y, coords, stride, x)


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
            if (FUSE_BLOCKS and isinstance(m, spnn.Conv3d) and isinstance(nxt, spnn.BatchNorm) and nxt.training
                    and nxt.fused_relu and not torch.is_tensor(x)
                    and not (i + 2 == len(mods) and residual is not None)):
                y = fused_conv_norm_act(x, m, nxt)          # training: one autograd node
                if y is not None:
                    x = y
                    i += 2
                    continue
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

    def _fused(self, x):
        """The whole block as one autograd node (training, standard configuration), else None."""
        net = self.net
        c1, n1, c2, n2 = net[0], net[1], net[3], net[4]
        if not (FUSE_BLOCKS and B.wants_grad(x.F, c1.kernel) and _plain_conv(c1) and _plain_conv(c2)
                and not c1.transposed and not c2.transposed and c1.stride == (1, 1, 1) and c2.stride == (1, 1, 1)
                and _bn_ok(n1, x.F) and _bn_ok(n2, x.F) and n1.fused_relu and not n2.fused_relu):
            return None
        ks = gs = bs = ns = None
        if not isinstance(self.downsample, nn.Identity):
            cs, ns = self.downsample[0], self.downsample[1]
            if not (cs.kernel_size == (1, 1, 1) and cs.stride == (1, 1, 1) and cs.bias is None and _bn_ok(ns, x.F)
                    and not ns.fused_relu):
                return None
            ks, gs, bs = cs.kernel, ns.weight, ns.bias
        elif x.F.shape[1] != c2.out_channels:
            return None
        kmap, coords, stride = _kmap_of(x, c1)
        stats = (bool(c1.bn_follows), bool(c2.bn_follows), bool(ks is not None and self.downsample[0].bn_follows))
        out = _Residual.apply(x.F, c1.kernel, n1.weight, n1.bias, c2.kernel, n2.weight, n2.bias, ks, gs, bs, kmap,
                              n1, n2, ns, stats)
        return _wrap(out, coords, stride, x)

    def forward(self, x):
        y = self._fused(x)
        if y is not None:
            return y
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
