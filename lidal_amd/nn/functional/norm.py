"""Batch normalisation over the rows of a [N, C] feature matrix on the HIP backend
(lidal_bn_* in include/lidal_amd.h).  Same arithmetic contract as torch.nn.functional.batch_norm:
biased batch variance for normalisation, running statistics updated with the unbiased variance."""
import os

import torch
from torch.autograd import Function

from ... import backend as B

__all__ = ['batch_norm_rows', 'column_sum', 'supported', 'train_forward', 'train_backward']


def supported(x, weight, bias):
    if not x.is_cuda or x.dim() != 2 or weight is None or bias is None:
        return False
    if x.dtype == torch.float32:
        vec = 4
    elif x.dtype == torch.bfloat16:
        vec = 8
    else:
        return False
    c = x.shape[1]
    return x.shape[0] > 1 and c % vec == 0 and c // vec <= 256 and weight.dtype == torch.float32


def _ws(n, c, dev):
    nbytes = B.lib().lidal_bn_workspace_bytes(n, c)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def train_forward(x, weight, bias, running_mean, running_var, momentum, eps, relu, num_batches_tracked=None,
                  tile_stats=None, residual=None, relu_after=False):
    """The training forward on raw tensors (no autograd): -> (y, save_mean, save_invstd, x contiguous, w, b).
    `relu` / `residual` / `relu_after` as in batch_norm_rows.  Shared by BatchNormRows and the fused
    block Functions of lidal_amd.network (which keep the saved tensors themselves)."""
    x = x.contiguous()
    if residual is not None:        # y = act(bn(x)) + residual in the normalising pass
        assert residual.shape == x.shape
        residual = residual.contiguous().to(x.dtype)
    relu = int(bool(relu)) | (2 if (relu_after and residual is not None) else 0)
    n, c = x.shape
    dev = x.device
    y = torch.empty_like(x)
    # (inside an autograd Function nothing is tracked: the parameters themselves, no detached copies)
    w = weight if weight.is_contiguous() else weight.detach().contiguous()
    b = bias if bias.is_contiguous() else bias.detach().contiguous()
    code = B.dtype_code(x.dtype)
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    invstd = torch.empty(c, dtype=torch.float32, device=dev)
    if tile_stats is not None and tile_stats.shape != (-(-n // B.stats_tile_rows()), c, 3):
        tile_stats = None
    if tile_stats is not None:      # statistics came with x from the producing convolution
        B.check(B.lib().lidal_bn_train_fwd_tiles(B.ptr(x), code, n, c, B.ptr(w), B.ptr(b), float(eps),
                                                 float(momentum), B.ptr(running_mean),
                                                 B.ptr(running_var), B.ptr(num_batches_tracked),
                                                 int(relu), B.ptr(residual), B.ptr(y), B.ptr(mean),
                                                 B.ptr(invstd), B.ptr(tile_stats),
                                                 tile_stats.shape[0], B.stream()),
                'bn_train_fwd')
    else:
        ws, nbytes = _ws(n, c, dev)
        B.check(B.lib().lidal_bn_train_fwd(B.ptr(x), code, n, c, B.ptr(w), B.ptr(b), float(eps),
                                           float(momentum), B.ptr(running_mean),
                                           B.ptr(running_var), B.ptr(num_batches_tracked), int(relu),
                                           B.ptr(residual), B.ptr(y), B.ptr(mean),
                                           B.ptr(invstd), B.ptr(ws), nbytes, B.stream()),
                'bn_train_fwd')
    # the kernels wrote the running statistics through raw pointers: move their version counters,
    # which key the folded eval-mode maps (network/blocks.py _fold) -- a train-mode forward
    # that no backward pass follows (recalibration under no_grad) must invalidate them too
    for t in (running_mean, running_var):
        if t is not None:
            torch.autograd.graph.increment_version(t)
    return y, mean, invstd, x, w, b


def train_backward(x, w, b, mean, invstd, relu, grad_out, need_dx=True, mask_from=None, tile_sums=None):
    """The training backward on raw tensors: -> (dx or None, grad_gamma f32 [C], grad_beta f32 [C], dy).
    `mask_from` (the block output y = relu(bn(x) + residual)): grad_out is first masked where y <= 0;
    the masked gradient `dy` (what also flows to the residual) is returned as the 4th value.
    `tile_sums` (f32 [tiles, C, 2], left on grad_out by the data-gradient launch that produced it,
    conv.conv_backward(bnb=...)): the backward sums per tile -- no pass over (x, dy) for them."""
    n, c = x.shape
    if mask_from is not None:       # relu(bn(x) + residual): dy where the output is positive, for both
        g0 = grad_out.contiguous().to(mask_from.dtype)
        grad_out = torch.empty_like(mask_from)
        B.check(B.lib().lidal_add_relu_bwd(B.ptr(mask_from), B.ptr(g0), B.ptr(grad_out), mask_from.numel(),
                                           B.dtype_code(mask_from.dtype), B.stream()), 'add_relu_bwd')
    # a channel slice of a concatenation's gradient (up stages) is read in place by the kernels
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if (grad_out.dim() == 2 and grad_out.dtype == x.dtype and grad_out.stride(1) == 1
            and grad_out.stride(0) >= c and grad_out.stride(0) % vec == 0
            and grad_out.storage_offset() % vec == 0):
        g = grad_out
    else:
        g = grad_out.contiguous().to(x.dtype)
    dx = torch.empty_like(x) if need_dx else None
    gg = torch.empty(c, dtype=torch.float32, device=x.device)
    gb = torch.empty(c, dtype=torch.float32, device=x.device)
    if (tile_sums is not None and mask_from is None
            and tuple(tile_sums.shape) == (-(-n // B.stats_tile_rows()), c, 2)):
        B.check(B.lib().lidal_bn_bwd_tiles(B.ptr(x), B.ptr(g), g.stride(0), B.dtype_code(x.dtype), n, c, B.ptr(w),
                                           B.ptr(b), int(relu), B.ptr(mean), B.ptr(invstd), B.ptr(dx),
                                           B.ptr(gg), B.ptr(gb), B.ptr(tile_sums), tile_sums.shape[0],
                                           B.stream()), 'bn_bwd')
        B.hit('bn_bwd(tile sums)')
        return dx, gg, gb, grad_out
    ws, nbytes = _ws(n, c, x.device)
    B.check(B.lib().lidal_bn_bwd(B.ptr(x), B.ptr(g), g.stride(0), B.dtype_code(x.dtype), n, c, B.ptr(w),
                                 B.ptr(b), int(relu), B.ptr(mean), B.ptr(invstd), B.ptr(dx),
                                 B.ptr(gg), B.ptr(gb),
                                 B.ptr(ws), nbytes, B.stream()), 'bn_bwd')
    return dx, gg, gb, grad_out


# The tail of a residual block backwards in fewer passes: the ReLU mask of relu(bn2(.) + shortcut) and the first pass of
# the BatchNorm backward of bn2 (and of the shortcut's BatchNorm) in ONE launch (lidal_add_relu_bwd_bn_sums), then
# lidal_bn_bwd_from_sums (merge + dx) per BatchNorm -- instead of lidal_add_relu_bwd and one lidal_bn_bwd each.  The partial
# sums are those of the separate pass bit for bit (tests/test_ops_gpu.py).  One pass over 4-5 arrays instead of 3 + 2
# (+ 2): at most a quarter of the tail's bytes, and the reducing kernel (256 workgroups, f64 sums) streams slower than
# the element-wise mask it absorbs -- measured (scripts/gpu/archive/tail_ab.sh, same box): one scan 6.74 / 6.75 -> 6.63 / 6.65 ms
# (23 launches fewer), 5 scans 15.18 / 15.20 -> 15.39 / 15.42 ms with every level fused.  Hence the row limit: the levels
# where a launch costs more than its bytes.  LIDAL_TAIL_SUMS_ROWS=0: the separate passes everywhere.
# Round 5: OFF by default (row limit 0).  bf16 takes the element-wise form below on every level; what was left to this
# pair was the f32 parity mode, and there it turned out to be the one place where a training run is not bit-reproducible:
# with the weight gradients running beside it on their side stream, 4 of 5 rounds of 16 x 6 SPVCNN f32 steps showed one
# repetition whose gradients differ from the others in the last bit (scripts/exp/determinism_steps.py, scripts/gpu/archive/r5_flake2.sh;
# the same rate at the end of round 4) -- none in 3 rounds with the separate passes, none without the side stream.  The
# kernels agree bit for bit whenever they run alone (tests/test_ops_gpu.py); the cause is not found.  LIDAL_TAIL_SUMS_ROWS=100000
# restores the round-4 behaviour.
TAIL_SUMS_ROWS = int(os.environ.get('LIDAL_TAIL_SUMS_ROWS', '0'))


def tail_sums(n):
    """Does the tail of a residual block of n rows run as the fused pair of launches?  (One rule for the per-operator
    path and the planned step: they must stay bitwise equal.)"""
    return 0 < n < TAIL_SUMS_ROWS


# bf16 (round 5): the mask as an element-wise launch that leaves f32 sums per slab of rows in the layout of the
# convolutions' tile sums (lidal_add_relu_bwd_bn_tile_sums), merged by lidal_bn_bwd_tiles in front of its dx pass -- 11
# passes over the level's arrays instead of 13 (a block with a shortcut BatchNorm), 7 instead of 8 (without) against the
# separate passes, and an element-wise kernel on 512 workgroups instead of the f64 reduction on 256 against the fused
# pair above.  Measured on every level (scripts/gpu/archive/r5_tail_rows.sh, row limits 100 000 / 30 000 / 1): 5 scans 13.99-14.07
# / 13.98-14.00 / 13.93-13.95 ms, one scan 6.30-6.55 / 6.25-6.40 / 6.18 ms -> every level.  The f32 parity mode keeps
# the f64 sums its golden gradients were taken with.  LIDAL_TAIL_TILES=0: off; LIDAL_TAIL_TILES_ROWS: from that many rows.
TAIL_TILES = os.environ.get('LIDAL_TAIL_TILES', '1') != '0'
TAIL_TILES_ROWS = int(os.environ.get('LIDAL_TAIL_TILES_ROWS', '1'))


def tail_tiles(n, dtype):
    """Does the tail of a residual block of n rows run as the element-wise mask with slab sums?  (One rule for the
    per-operator path and the planned step.)"""
    return TAIL_TILES and n >= TAIL_TILES_ROWS > 0 and dtype == torch.bfloat16


def _tail_backward_tiles(grad_out, out, x2, w2, b2, mean2, inv2, shortcut):
    n, c = x2.shape
    dev = x2.device
    code = B.dtype_code(x2.dtype)
    L = B.lib()
    g0 = grad_out.contiguous().to(out.dtype)
    gm = torch.empty_like(out)
    parts = int(L.lidal_bn_tail_parts(n, c, code))
    sums2 = torch.empty((c, parts, 2), dtype=torch.float32, device=dev)
    sumss = torch.empty((c, parts, 2), dtype=torch.float32, device=dev) if shortcut is not None else None
    xs, ws, bs, means, invs = shortcut if shortcut is not None else (None,) * 5
    B.check(L.lidal_add_relu_bwd_bn_tile_sums(B.ptr(out), B.ptr(g0), B.ptr(gm), code, n, c, B.ptr(x2), B.ptr(mean2),
                                              B.ptr(inv2), B.ptr(sums2), B.ptr(xs), B.ptr(means), B.ptr(invs),
                                              B.ptr(sumss), parts, B.stream()), 'add_relu_bwd')
    res = []
    for x, w, b, mu, inv, sums in ((x2, w2, b2, mean2, inv2, sums2), (xs, ws, bs, means, invs, sumss))[:2 if shortcut is not None else 1]:
        dx = torch.empty_like(x)
        gg = torch.empty(c, dtype=torch.float32, device=dev)
        gb = torch.empty(c, dtype=torch.float32, device=dev)
        B.check(L.lidal_bn_bwd_tiles(B.ptr(x), B.ptr(gm), c, code, n, c, B.ptr(w), B.ptr(b), 0, B.ptr(mu), B.ptr(inv),
                                     B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(sums), parts, B.stream()), 'bn_bwd')
        B.hit('bn_bwd(tile sums)')
        res.append((dx, gg, gb))
    return gm, res[0], (res[1] if shortcut is not None else None)


def tail_backward(grad_out, out, x2, w2, b2, mean2, inv2, shortcut=None):
    """Backward of out = relu(bn2(x2) + s) (network/utils.py:142-172), s = the identity branch or bn_s(xs) with
    shortcut = (xs, ws, bs, means, invs): -> (gm, (dx2, gg2, gb2), (dxs, ggs, gbs) or None); gm = the masked gradient,
    which is also the identity branch's."""
    if tail_tiles(x2.shape[0], x2.dtype) and x2.is_cuda:
        return _tail_backward_tiles(grad_out, out, x2, w2, b2, mean2, inv2, shortcut)
    if not tail_sums(x2.shape[0]):
        dx2, gg2, gb2, gm = train_backward(x2, w2, b2, mean2, inv2, False, grad_out, True, out)
        side = None
        if shortcut is not None:
            xs, ws, bs, means, invs = shortcut
            side = train_backward(xs, ws, bs, means, invs, False, gm, True)[:3]
        return gm, (dx2, gg2, gb2), side
    n, c = x2.shape
    dev = x2.device
    code = B.dtype_code(x2.dtype)
    g0 = grad_out.contiguous().to(out.dtype)
    gm = torch.empty_like(out)
    nbytes = B.lib().lidal_bn_workspace_bytes(n, c)
    part2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    parts = torch.empty(nbytes, dtype=torch.uint8, device=dev) if shortcut is not None else None
    xs, ws, bs, means, invs = shortcut if shortcut is not None else (None,) * 5
    B.check(B.lib().lidal_add_relu_bwd_bn_sums(B.ptr(out), B.ptr(g0), B.ptr(gm), code, n, c, B.ptr(x2), B.ptr(mean2),
                                               B.ptr(inv2), B.ptr(part2), B.ptr(xs), B.ptr(means), B.ptr(invs),
                                               B.ptr(parts), nbytes, B.stream()), 'add_relu_bwd')
    res = []
    for x, w, b, mu, inv, part in ((x2, w2, b2, mean2, inv2, part2), (xs, ws, bs, means, invs, parts))[:2 if shortcut is not None else 1]:
        dx = torch.empty_like(x)
        gg = torch.empty(c, dtype=torch.float32, device=dev)
        gb = torch.empty(c, dtype=torch.float32, device=dev)
        B.check(B.lib().lidal_bn_bwd_from_sums(B.ptr(x), B.ptr(gm), c, code, n, c, B.ptr(w), B.ptr(b), 0, B.ptr(mu),
                                               B.ptr(inv), B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(part), nbytes,
                                               B.stream()), 'bn_bwd')
        res.append((dx, gg, gb))
    return gm, res[0], (res[1] if shortcut is not None else None)


class BatchNormRows(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu,
                num_batches_tracked=None, tile_stats=None, residual=None, relu_after=False):
        if residual is not None:
            assert training
        relu_bits = int(bool(relu)) | (2 if (relu_after and residual is not None) else 0)
        if training:
            y, mean, invstd, x, w, b = train_forward(x, weight, bias, running_mean, running_var, momentum, eps,
                                                     relu, num_batches_tracked, tile_stats, residual, relu_after)
        else:
            x = x.contiguous()
            y = torch.empty_like(x)
            w, b = weight.detach().contiguous(), bias.detach().contiguous()
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
            B.check(B.lib().lidal_bn_eval_fwd(B.ptr(x), B.dtype_code(x.dtype), x.shape[0], x.shape[1], B.ptr(w),
                                              B.ptr(b), B.ptr(running_mean), B.ptr(running_var), float(eps),
                                              int(relu_bits), B.ptr(y), B.stream()), 'bn_eval_fwd')
        ctx.training = training
        ctx.relu = bool(relu_bits & 1)
        ctx.relu_after = bool(relu_bits & 2)
        ctx.has_residual = residual is not None
        if ctx.relu_after:              # the output is the mask of the trailing ReLU
            ctx.save_for_backward(x, w, b, mean, invstd, y)
        else:
            ctx.save_for_backward(x, w, b, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        B.note_backward()
        x, w, b, mean, invstd = ctx.saved_tensors[:5]
        if not ctx.training:            # eval-mode backward (not on the LiDAL path): plain torch
            g = grad_out.contiguous().to(x.dtype)
            xhat = (x.float() - mean) * invstd
            gf = g.float()
            if ctx.relu:
                gf = gf * ((xhat * w + b) > 0)
            return ((gf * (w * invstd)).to(x.dtype), (gf * xhat).sum(0), gf.sum(0), None, None,
                    None, None, None, None, None, None, None, None)
        dx, gg, gb, dy = train_backward(x, w, b, mean, invstd, ctx.relu, grad_out, ctx.needs_input_grad[0],
                                        ctx.saved_tensors[5] if ctx.relu_after else None)
        # the residual passes straight through: its gradient is grad_out itself (masked, if a ReLU followed)
        return (dx, gg, gb, None, None, None, None, None, None, None, None,
                dy if ctx.has_residual and ctx.needs_input_grad[11] else None, None)


def batch_norm_rows(x, weight, bias, running_mean, running_var, training, momentum, eps,
                    relu=False, num_batches_tracked=None, tile_stats=None, residual=None, relu_after=False):
    """`num_batches_tracked` (i64 scalar buffer, training only) is incremented inside the kernel.
    `tile_stats` (training only): f32 [ceil(N/128), C, 3] (count, mean, M2) per 128-row tile, written
    by the convolution that produced x (conv3d(..., want_stats=True)): no statistics pass over x.
    `residual` (training only, [N, C]): returns act(bn(x)) + residual from the same pass, with a ReLU
    on the sum if `relu_after` (the end of a residual block)."""
    if tile_stats is not None and not (training and tile_stats.shape == (-(-x.shape[0] // B.stats_tile_rows()), x.shape[1], 3)):
        tile_stats = None
    if not training and not B.wants_grad(x, weight, bias):      # inference: one kernel, no node
        x = x.contiguous()
        y = torch.empty_like(x)
        B.check(B.lib().lidal_bn_eval_fwd(B.ptr(x), B.dtype_code(x.dtype), x.shape[0], x.shape[1],
                                          B.ptr(weight), B.ptr(bias), B.ptr(running_mean),
                                          B.ptr(running_var), float(eps), int(relu), B.ptr(y),
                                          B.stream()), 'bn_eval_fwd')
        return y
    if residual is not None and not training:
        y = batch_norm_rows(x, weight, bias, running_mean, running_var, training, momentum, eps, relu,
                            num_batches_tracked, tile_stats) + residual
        return torch.relu(y) if relu_after else y
    return BatchNormRows.apply(x, weight, bias, running_mean, running_var, training, momentum, eps,
                               relu, num_batches_tracked if training else None, tile_stats, residual,
                               relu_after)


def column_sum(x):
    """f32 [C] column sums of x [N, C] (f32 / bf16), hierarchical and reproducible."""
    x = x.contiguous()
    n, c = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = B.lib().lidal_bn_workspace_bytes(n, c) + 12 * c
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    B.check(B.lib().lidal_colsum(B.ptr(x), B.dtype_code(x.dtype), n, c, B.ptr(out), B.ptr(ws),
                                 nbytes, B.stream()), 'colsum')
    return out
