"""F.spdevoxelize and F.calc_ti_weights (torchsparse/nn/functional/devoxelize.py;
network/utils.py:77-83,95): 8-corner trilinear voxel -> point interpolation."""
import torch
from torch.autograd import Function

from ... import backend as B
from .invlist import inverse_lists, segment_workspace

import os

__all__ = ['spdevoxelize', 'calc_ti_weights', 'ti_weights_and_index', 'prepare_devoxelize', 'devox_cells', 'cells_mode']

# devoxelize backward through the cells (csrc/voxel.hip devox_cell_sums_kernel): where a voxel of the level holds at least
# this many points on average (stride 16: ~24; stride 4: ~4, where the per-voxel lists are the shorter way).  0 = never.
CELLS_MIN_AVG = int(os.environ.get('LIDAL_DEVOX_CELLS_AVG', '12'))


def cells_mode(idx8, n_points, m, c, dtype=None):
    """Does the backward of spdevoxelize over (idx8 [N, 8], m voxels, c channels) run through the cells?  Only for an
    index that network/glue.py corner_tables built FOR A CALLER WHO VOUCHED that every point's own voxel exists
    (own_cells: it marks the index; every point of a cell -- idx8[:, 0], the voxel its own coordinates floor to -- then
    has the same eight corners, utils.py:67-79, and no point is without a cell), on levels with many points per voxel.
    One rule for the per-operator path and the planned step."""
    if not CELLS_MIN_AVG or not getattr(idx8, '_lidal_cell_corners', False) or m <= 0:
        return False
    # (whole power-of-two counts of 16-byte steps per row under both element types)
    return n_points >= CELLS_MIN_AVG * m and c in (32, 64, 128, 256)


def devox_cells(idx8, m):
    """(vorder i32 [N], vseg i64 [m + 1], corder i32 [8 m], cseg i64 [m + 1]): the points of every cell and, for every
    voxel, the (cell * 8 + corner) entries it is a corner of.  Cached on idx8."""
    key = (m, idx8._version, idx8.data_ptr(), idx8.numel())
    cached = getattr(idx8, '_lidal_cells', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    # the points' own voxel index: idx8[:, 0] -- or the tensor F.spvoxelize already keeps lists for (the same values:
    # both are queries of the floored coordinates, utils.py:44-52 / 72-77; network/geometry.py links them)
    cell = getattr(idx8, '_lidal_cell_index', None)
    if cell is None or cell.shape[0] != idx8.shape[0]:
        cell = idx8[:, 0].contiguous()
    vorder, vseg = inverse_lists(cell, m)
    # (every voxel of the level holds a point; a voxel without one would take a neighbour's corners for sums that stay 0)
    first = vorder[vseg[:-1].clamp(max=max(int(vorder.shape[0]) - 1, 0))].long()
    cidx = idx8[first].contiguous()                    # [m, 8]: the corners of every cell
    corder, cseg = inverse_lists(cidx, m)
    out = (vorder, vseg, corder, cseg)
    idx8._lidal_cells = (key, out, cell, cidx)
    return out



def ti_weights_and_index(coords, idx_query, scale=1):
    """Fused form used by lidal_amd.network: returns (w f32 [N,8], idx i32 [N,8]), i.e.
    calc_ti_weights(...).transpose(0,1) and idx_query.transpose(0,1) of network/utils.py:77-79
    in one pass."""
    B.require_gpu(coords, idx_query)
    coords = coords.contiguous().float()
    idx_query = idx_query.contiguous()
    assert idx_query.dtype == torch.int64 and idx_query.shape[0] == 8
    n = coords.shape[0]
    w = B.empty((n, 8), torch.float32, coords.device)
    idx32 = B.empty((n, 8), torch.int, coords.device)
    B.check(B.lib().lidal_ti_weights(B.ptr(coords), coords.shape[1], B.ptr(idx_query), n,
                                     float(scale), B.ptr(w), B.ptr(idx32), B.stream()),
            'ti_weights')
    return w, idx32


def calc_ti_weights(coords, idx_query, scale=1):
    """torchsparse signature: returns w [8, N] (the caller transposes)."""
    w, _ = ti_weights_and_index(coords, idx_query, scale)
    return w.transpose(0, 1).contiguous()


class DevoxelizeFunction(Function):
    @staticmethod
    def forward(ctx, feats, coords, weights):
        B.require_gpu(feats, coords, weights)
        in_dtype = feats.dtype
        # bf16 rows travel as bf16 (f32 accumulation inside the kernel); others on an f32 copy
        native = in_dtype == torch.bfloat16
        feats = feats.contiguous() if native else feats.contiguous().float()
        if coords.dtype != torch.int or not coords.is_contiguous():
            coords = coords.contiguous().int()
        if weights.dtype != torch.float32 or not weights.is_contiguous():
            weights = weights.contiguous().float()
        m, c = feats.shape
        n = coords.shape[0]
        out = torch.empty((n, c), dtype=feats.dtype, device=feats.device)
        B.check(B.lib().lidal_devoxelize_fwd(B.ptr(feats), B.ptr(coords), B.ptr(weights),
                                             B.ptr(out), n, m, c, B.dtype_code(feats.dtype),
                                             B.stream()), 'devoxelize_fwd')
        ctx.for_backwards = (coords, weights, m, in_dtype)
        return out.to(in_dtype)

    @staticmethod
    def backward(ctx, grad_output):
        coords, weights, m, in_dtype = ctx.for_backwards
        native = (in_dtype == torch.bfloat16 and grad_output.dtype == torch.bfloat16
                  and grad_output.shape[1] % 4 == 0)
        g = grad_output.contiguous() if native else grad_output.contiguous().float()
        n, c = g.shape
        gin = torch.empty((m, c), dtype=g.dtype, device=g.device)
        if cells_mode(coords, n, m, c, g.dtype):        # the coarse levels: every gradient row read once
            vorder, vseg, corder, cseg = devox_cells(coords, m)
            nbytes = B.lib().lidal_devoxelize_bwd_cells_workspace_bytes(m, c)
            ws = B.workspace(nbytes, g.device)
            B.check(B.lib().lidal_devoxelize_bwd_cells(B.ptr(g), B.ptr(vorder), B.ptr(vseg), B.ptr(weights), B.ptr(corder),
                                                       B.ptr(cseg), B.ptr(gin), m, c, B.dtype_code(g.dtype), B.ptr(ws),
                                                       nbytes, B.stream()), 'devoxelize_bwd_cells')
        elif c % 4 == 0:    # ordered per-voxel gather: no atomics, reproducible
            order, seg_ptr = inverse_lists(coords, m, weights)
            ws, nbytes = segment_workspace(8 * n, m, c, g.device)
            B.check(B.lib().lidal_devoxelize_bwd_sorted(B.ptr(g), B.ptr(order), B.ptr(seg_ptr),
                                                        B.ptr(weights), B.ptr(gin), m, c,
                                                        B.dtype_code(g.dtype), 8 * n, B.ptr(ws),
                                                        nbytes, B.stream()),
                    'devoxelize_bwd_sorted')
        else:
            B.check(B.lib().lidal_devoxelize_bwd(B.ptr(g), B.ptr(coords), B.ptr(weights),
                                                 B.ptr(gin), n, m, c, B.stream()), 'devoxelize_bwd')
        return gin.to(in_dtype), None, None


def prepare_devoxelize(coords, weights, m, c=None):
    """What the backward of F.spdevoxelize derives from index and weights alone (network/geometry.py): the
    per-voxel contributor lists of the ordered scatter sum.  coords i32 [N,8], weights f32 [N,8], m voxels."""
    if coords.dtype == torch.int and coords.is_contiguous() and weights.dtype == torch.float32 \
            and weights.is_contiguous():
        if c is not None and cells_mode(coords, coords.shape[0], m, c):
            devox_cells(coords, m)
        else:
            inverse_lists(coords, m, weights)


def spdevoxelize(feats, coords, weights):
    if B.wants_grad(feats):
        return DevoxelizeFunction.apply(feats, coords, weights)
    return DevoxelizeFunction.forward(B.NoGradCtx(), feats, coords, weights)        # inference: no autograd node
