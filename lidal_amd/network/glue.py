"""Point <-> voxel exchange of SPVCNN (counterparts of /root/reference/network/utils.py:13-102),
composed from the HIP operators.  Results (indices, weights, features) follow the reference
expression by expression; the caches live on the PointTensor exactly as upstream keeps them so
the stride-1 tables built after the stem are re-used by the last up-stage.
"""
import torch

from .. import PointTensor, SparseTensor
from .. import backend as B
from ..nn import functional as F
from ..nn.utils import get_kernel_offsets

__all__ = ['initial_voxelize', 'point_to_voxel', 'voxel_to_point', 'initial_tables', 'point_tables', 'corner_tables']



RENUMBER = None         # experiments only: (idx_query, counts, coords) -> the same three under another level-0 voxel order


def _floor_to_stride(z, s):
    """(floor(xyz / s).int() * s, batch.int()) as one int32 [N,4] tensor (utils.py:44-47,72-75)."""
    c = z.C
    if c.is_cuda and c.dtype == torch.float32 and c.dim() == 2 and c.shape[1] == 4 and c.is_contiguous() \
            and float(s) == int(s):
        out = B.empty(c.shape, torch.int32, c.device)      # one pass (csrc/hash.hip)
        B.check(B.lib().lidal_floor_coords(B.ptr(c), c.shape[0], int(s), B.ptr(out), B.stream()), 'floor_coords')
        return out
    xyz = torch.floor(c[:, :3] / s).int() * s
    return torch.cat([xyz, c[:, -1].int().view(-1, 1)], 1)


def initial_tables(z, init_res, after_res):
    """The coordinate half of initial_voxelize (utils.py:14-22,28-31): point -> voxel index, counts and the voxel
    coordinates, left in z's caches; returns the coordinates.  Also what network/geometry.py runs ahead of the
    features."""
    # true IEEE division: torch's GPU `tensor / python_scalar` multiplies by the reciprocal, which
    # leaves voxel centres 1e-7 off the integers the CPU path produces
    c = z.C
    if c.is_cuda and c.dtype == torch.float32 and c.dim() == 2 and c.shape[1] == 4 and c.is_contiguous():
        # one pass (csrc/hash.hip), the same separately rounded multiply and divide
        new_float_coord = B.empty(c.shape, c.dtype, c.device)
        floored_int = B.empty(c.shape, torch.int32, c.device)
        B.check(B.lib().lidal_revoxelize_coords(B.ptr(c), c.shape[0], float(init_res), float(after_res),
                                                B.ptr(new_float_coord), B.ptr(floored_int), B.stream()), 'revoxelize_coords')
        floored = None
    else:
        res = torch.full((), after_res, dtype=z.C.dtype, device=z.C.device)
        new_float_coord = torch.cat([(z.C[:, :3] * init_res) / res, z.C[:, -1].view(-1, 1)], 1)
        floored = torch.floor(new_float_coord)
        floored_int = floored.int()
    pc_hash = F.sphash(floored_int)
    sparse_hash = F.unique_sorted(pc_hash)
    idx_query = F.sphashquery(pc_hash, sparse_hash)
    counts = F.spcount(idx_query.int(), len(sparse_hash))
    if len(sparse_hash) == pc_hash.numel():
        # as many voxels as points (the reference's datasets voxelise the scans at this very resolution,
        # sk_dataset.py:160-171): the index is a permutation and every count is 1 -- F.spvoxelize then moves rows
        idx_query._lidal_one_to_one = True
    if getattr(idx_query, '_lidal_one_to_one', False):
        # mean of ONE row, rounded: the row itself -- the int rows permuted bit for bit (the 1:1 kernel copies 16-byte rows)
        inserted_coords = F.spvoxelize(floored_int.view(torch.float32), idx_query, counts).view(torch.int32)
    else:
        if floored is None:
            floored = floored_int.float()
        inserted_coords = torch.round(F.spvoxelize(floored, idx_query, counts)).int()
    if RENUMBER is not None:        # hook of scripts/exp/l0_morton.py (an experiment: the level-0 voxel order is internal)
        idx_query, counts, inserted_coords = RENUMBER(idx_query, counts, inserted_coords)
    z.additional_features['idx_query'][1] = idx_query
    z.additional_features['counts'][1] = counts
    z.additional_features['init_coords'] = inserted_coords     # the rows that index refers to
    z.C = new_float_coord
    return inserted_coords


def initial_voxelize(z, init_res, after_res):
    """utils.py:13-33: re-voxelise the points at `after_res`; the voxel order is the SORTED order
    of the distinct 60-bit coordinate hashes (torch.unique at :18)."""
    inserted_coords = initial_tables(z, init_res, after_res)
    inserted_feat = F.spvoxelize(z.F, z.additional_features['idx_query'][1], z.additional_features['counts'][1])
    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.cmaps.setdefault(new_tensor.stride, new_tensor.coords)
    return new_tensor


def point_tables(x, z):
    """The point -> voxel index and counts of point_to_voxel (utils.py:39-53) at x's stride, in z's caches."""
    cache_i, cache_c = z.additional_features['idx_query'], z.additional_features['counts']
    if cache_i.get(x.s) is None and tuple(x.s) == (1, 1, 1) and cache_i.get(1) is not None \
            and x.C is z.additional_features.get('init_coords'):
        # upstream recomputes here what initial_voxelize cached under the int key 1 (SURVEY.md
        # appendix A: "results identical"): same points, same voxel rows, so the same index and
        # counts -- and the same contributor lists, which are cached on the index tensor
        cache_i[x.s], cache_c[x.s] = cache_i[1], cache_c[1]
    if cache_i.get(x.s) is None:
        pc_hash = F.sphash(_floor_to_stride(z, x.s[0]))
        idx_query = F.coords_table(x.C, x.cmaps, x.s).query(pc_hash)     # == F.sphashquery(pc_hash, F.sphash(x.C))
        cache_i[x.s] = idx_query
        cache_c[x.s] = F.spcount(idx_query.int(), x.C.shape[0])
    _link_cells(z, x.s)
    return cache_i[x.s], cache_c[x.s]


def _link_cells(z, stride):
    """One list of the points of every voxel for both directions: the corner index of this stride (corner_tables) learns
    the point -> voxel index F.spvoxelize keeps lists for -- the same values as its own column 0 (F.devoxelize.devox_cells)."""
    idx8 = z.idx_query.get(stride)
    pidx = z.additional_features['idx_query'].get(stride)
    if idx8 is not None and pidx is not None and getattr(idx8, '_lidal_cell_index', None) is None \
            and pidx.shape[0] == idx8.shape[0]:
        from ..nn.functional.voxelize import _index32
        idx8._lidal_cell_index = _index32(pidx)


def point_to_voxel(x, z):
    """utils.py:38-61: mean of the point features falling into each voxel of x."""
    idx_query, counts = point_tables(x, z)
    # z.F has a second consumer downstream (the point-branch Linear of SPVCNN.forward): it gets an
    # alias whose gradient joins the voxelize backward in-kernel (F.spvoxelize, fork)
    if B.FORK & 2:
        feats, z.F = F.spvoxelize(z.F, idx_query, counts, fork=True)
    else:
        feats = F.spvoxelize(z.F, idx_query, counts)
    new_tensor = SparseTensor(feats, x.C, x.s)
    new_tensor.cmaps = x.cmaps
    new_tensor.kmaps = x.kmaps
    return new_tensor


def corner_tables(x, z, nearest=False, own_cells=False):
    """The 8-corner index and trilinear weights of voxel_to_point (utils.py:67-92) at x's stride, in z's caches.
    own_cells: the caller vouches that the voxel every point's own coordinates floor to EXISTS in x (x's coordinates
    derive from z: initial_voxelize and the levels below it, as in SPVCNN) -- only then may the backward of the
    devoxelisation run through the cells (F.devoxelize.cells_mode): a point whose own voxel is absent has no cell and
    would lose the gradient of its other seven corners there (the per-voxel lists keep it)."""
    if z.idx_query.get(x.s) is None or z.weights.get(x.s) is None:
        off = get_kernel_offsets(2, x.s, 1, device=z.C.device)
        old_hash = F.sphash(_floor_to_stride(z, x.s[0]), off)          # [8, N]
        idx_query = F.coords_table(x.C, x.cmaps, x.s).query(old_hash)    # == F.sphashquery(old_hash, F.sphash(x.C))
        weights, idx_query = F.ti_weights_and_index(z.C, idx_query, scale=x.s[0])   # [N,8] both
        if nearest:
            weights[:, 1:] = 0.
            idx_query[:, 1:] = -1
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
        _link_cells(z, x.s)
    if own_cells and not nearest:
        # every point of a cell (idx_query[:, 0]: the voxel its own coordinates floor to) has the same eight corners by
        # the expression above, and every point HAS a cell: what the cell form of the backward relies on
        z.idx_query[x.s]._lidal_cell_corners = True
    return z.idx_query[x.s], z.weights[x.s]


def voxel_to_point(x, z, nearest=False, own_cells=False):
    """utils.py:66-102: trilinear interpolation of the 8 surrounding voxels of x at each point.  own_cells: corner_tables."""
    idx_query, weights = corner_tables(x, z, nearest, own_cells)
    new_feat = F.spdevoxelize(x.F, idx_query, weights)
    new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
    new_tensor.additional_features = z.additional_features
    return new_tensor
