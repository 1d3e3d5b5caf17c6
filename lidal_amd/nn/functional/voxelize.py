"""F.spvoxelize (torchsparse/nn/functional/voxelize.py; network/utils.py:22,25,56):
mean-pool point rows into voxel rows, with the matching backward."""
import torch
from torch.autograd import Function

from ... import backend as B
from .invlist import inverse_lists, segment_workspace

__all__ = ['spvoxelize', 'prepare_voxelize']


def _index32(coords):
    """int32 copy of a point -> voxel index, cached on the index tensor (keyed on its version counter and storage)."""
    key = (coords._version, coords.data_ptr(), coords.numel())
    cached = getattr(coords, '_lidal_i32', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    idx32 = coords.contiguous().int()
    coords._lidal_i32 = (key, idx32)
    return idx32


def prepare_voxelize(coords, counts):
    """What F.spvoxelize derives from the index alone, ahead of the features (network/geometry.py): the int32 copy
    and, unless every voxel has one point, the per-voxel contributor lists of the ordered forward."""
    idx32 = _index32(coords)
    if not (getattr(coords, '_lidal_one_to_one', False) and coords.numel() == counts.shape[0]):
        inverse_lists(idx32, counts.shape[0])


class VoxelizeFunction(Function):
    @staticmethod
    def forward(ctx, feats, coords, counts, fork=False):
        B.require_gpu(feats, coords, counts)
        feats_in = feats
        in_dtype = feats.dtype
        # bf16 rows are read and written as bf16 (f32 accumulation inside the kernel) when the
        # ordered path applies; everything else computes on an f32 copy
        native = in_dtype == torch.bfloat16 and feats.shape[1] % 4 == 0
        feats = feats.contiguous() if native else feats.contiguous().float()
        idx32 = _index32(coords)
        counts = counts.contiguous().int()
        n, c = feats.shape
        m = counts.shape[0]
        out = torch.empty((m, c), dtype=feats.dtype, device=feats.device)
        if getattr(coords, '_lidal_one_to_one', False) and n == m:
            # one point per voxel (network/glue.py initial_voxelize found as many voxels as points): the mean is
            # the row itself -- a row permutation instead of contributor lists + one wave per voxel
            B.check(B.lib().lidal_voxelize_fwd_1to1(B.ptr(feats), B.ptr(idx32), B.ptr(out), n, c,
                                                    B.dtype_code(feats.dtype), B.stream()), 'voxelize_fwd_1to1')
        elif c % 4 == 0:    # ordered per-voxel gather: no atomics, reproducible
            order, seg_ptr = inverse_lists(idx32, m)
            ws, nbytes = segment_workspace(n, m, c, feats.device)
            B.check(B.lib().lidal_voxelize_fwd_sorted(B.ptr(feats), B.ptr(order), B.ptr(seg_ptr),
                                                      B.ptr(counts), B.ptr(out), m, c,
                                                      B.dtype_code(feats.dtype), n, B.ptr(ws),
                                                      nbytes, B.stream()),
                    'voxelize_fwd_sorted')
        else:
            B.check(B.lib().lidal_voxelize_fwd(B.ptr(feats), B.ptr(idx32), B.ptr(counts),
                                               B.ptr(out), n, m, c, B.stream()), 'voxelize_fwd')
        coords = idx32
        ctx.for_backwards = (coords, counts, n, in_dtype)
        # fork: an alias of the input for its second consumer; the gradient arriving through it is
        # added inside the backward kernel (no separate accumulation pass over the point rows)
        return (out.to(in_dtype), feats_in) if fork else out.to(in_dtype)

    @staticmethod
    def backward(ctx, grad_output, grad_skip=None):
        coords, counts, n, in_dtype = ctx.for_backwards
        native = in_dtype == torch.bfloat16 and grad_output.dtype == torch.bfloat16
        g = grad_output.contiguous() if native else grad_output.contiguous().float()
        m, c = g.shape
        res = None if grad_skip is None else grad_skip.contiguous().to(g.dtype)
        gin = torch.empty((n, c), dtype=g.dtype, device=g.device)
        B.check(B.lib().lidal_voxelize_bwd(B.ptr(g), B.ptr(coords), B.ptr(counts), B.ptr(res), B.ptr(gin),
                                           n, m, c, B.dtype_code(g.dtype), B.stream()),
                'voxelize_bwd')
        return gin.to(in_dtype), None, None, None


def spvoxelize(feats, coords, counts, fork=False):
    """`fork`: returns (voxel rows, alias of `feats`) -- hand the alias to the other consumer of the
    point features; its gradient then joins this one's inside the backward kernel."""
    if B.wants_grad(feats):
        if fork and feats.requires_grad:
            return VoxelizeFunction.apply(feats, coords, counts, True)
        out = VoxelizeFunction.apply(feats, coords, counts)
    else:
        out = VoxelizeFunction.forward(B.NoGradCtx(), feats, coords, counts)     # inference: no autograd node
    return (out, feats) if fork else out
