"""CPU restatement of torchsparse 1.4.0 `torchsparse.nn.functional`.  TEST INFRASTRUCTURE.

Each function names the upstream file it follows (tag v1.4.0; the dependency is pinned at
/root/reference/docs/requirements.txt:191 but not vendored) and the reference call site that
constrains it.  Integer work is numpy uint64/int64 (bit-exact by construction); floating-point
work is plain differentiable torch so autograd supplies the backward oracle.
"""
import numpy as np
import torch

from .. import SparseTensor
from ..utils import make_ntuple
from .utils import get_kernel_offsets

__all__ = ['sphash', 'sphashquery', 'spcount', 'spvoxelize', 'spdevoxelize', 'calc_ti_weights',
           'spdownsample', 'conv3d', 'build_kmap']

_FNV_OFFSET = np.uint64(14695981039346656037)
_FNV_PRIME = np.uint64(1099511628211)
_MASK60 = np.uint64(0x0FFFFFFFFFFFFFFF)


def _fnv60(words):
    """words: uint64 [..., 4] (each a zero-extended uint32).  backend/hash/hash_cpu.cpp."""
    with np.errstate(over='ignore'):
        h = np.full(words.shape[:-1], _FNV_OFFSET, dtype=np.uint64)
        for j in range(4):
            h = h ^ words[..., j]
            h = h * _FNV_PRIME
        h = (h >> np.uint64(60)) ^ (h & _MASK60)
    return h.astype(np.int64)


def sphash(coords, offsets=None):
    """nn/functional/hash.py.  coords i32 [N,4] (x,y,z,b) -> i64 [N]; with offsets i32 [K,3]
    -> i64 [K,N] hashing (xyz+offset_k, b).  Call sites: network/utils.py:17,42-47,70-75."""
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    c = coords.detach().cpu().numpy().astype(np.int32)
    if offsets is None:
        w = c.view(np.uint32).astype(np.uint64)
        return torch.from_numpy(_fnv60(w))
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    o = offsets.detach().cpu().numpy().astype(np.int32)
    cur = np.broadcast_to(c[None, :, :], (o.shape[0],) + c.shape).copy()
    cur[:, :, :3] += o[:, None, :]          # int32 wrap-around add, as in C
    w = cur.view(np.uint32).astype(np.uint64)
    return torch.from_numpy(_fnv60(w))


def sphashquery(queries, references):
    """nn/functional/query.py + backend/hashmap/hashmap_cpu.hpp: position of each query in
    `references` or -1; the first occurrence wins for duplicate reference keys
    (dense_hash_map::insert keeps the existing entry).  Call sites: network/utils.py:19,48,76."""
    sizes = queries.size()
    q = queries.detach().cpu().numpy().reshape(-1).astype(np.int64)
    r = references.detach().cpu().numpy().reshape(-1).astype(np.int64)
    if r.size == 0:
        return torch.full(sizes, -1, dtype=torch.long)
    uniq, first = np.unique(r, return_index=True)
    pos = np.searchsorted(uniq, q)
    pos[pos >= uniq.size] = uniq.size - 1
    hit = uniq[pos] == q
    out = np.where(hit, first[pos], -1).astype(np.int64)
    return torch.from_numpy(out).view(*sizes)


def spcount(coords, num):
    """nn/functional/count.py + backend/others/count_cpu.cpp: int32 histogram of the
    non-negative entries.  Call sites: network/utils.py:20,49."""
    idx = coords.detach().cpu().numpy().astype(np.int64)
    idx = idx[idx >= 0]
    return torch.from_numpy(np.bincount(idx, minlength=num)[:num].astype(np.int32))


def spvoxelize(feats, coords, counts):
    """nn/functional/voxelize.py + backend/voxelize: out[idx[i]] += feats[i] / counts[idx[i]]
    (mean pool; idx<0 skipped).  Differentiable.  Call sites: network/utils.py:22,25,56."""
    idx = coords.long()
    valid = idx >= 0
    idx_v = idx[valid]
    contrib = feats[valid] / counts[idx_v].to(feats.dtype).unsqueeze(1)
    out = torch.zeros(counts.shape[0], feats.shape[1], dtype=feats.dtype)
    return out.index_add(0, idx_v, contrib)


def spdevoxelize(feats, coords, weights):
    """nn/functional/devoxelize.py + backend/devoxelize: out[i] = sum_k w[i,k]*feats[idx[i,k]]
    with idx -1 skipped.  Differentiable.  Call sites: network/utils.py:83,95."""
    idx = coords.long()
    out = torch.zeros(idx.shape[0], feats.shape[1], dtype=feats.dtype)
    for k in range(idx.shape[1]):
        ik = idx[:, k]
        valid = ik >= 0
        rows = feats[ik.clamp(min=0)] * weights[:, k:k + 1]
        out = out + torch.where(valid.unsqueeze(1), rows, torch.zeros_like(rows))
    return out


def calc_ti_weights(coords, idx_query, scale=1):
    """nn/functional/devoxelize.py::calc_ti_weights -> [8,N].  Call site: network/utils.py:77."""
    with torch.no_grad():
        p = coords
        if scale != 1:
            pf = torch.floor(coords / scale) * scale
        else:
            pf = torch.floor(coords)
        pc = pf + scale
        x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
        xf, yf, zf = pf[:, 0:1], pf[:, 1:2], pf[:, 2:3]
        xc, yc, zc = pc[:, 0:1], pc[:, 1:2], pc[:, 2:3]
        w0 = (xc - x) * (yc - y) * (zc - z)
        w1 = (xc - x) * (yc - y) * (z - zf)
        w2 = (xc - x) * (y - yf) * (zc - z)
        w3 = (xc - x) * (y - yf) * (z - zf)
        w4 = (x - xf) * (yc - y) * (zc - z)
        w5 = (x - xf) * (yc - y) * (z - zf)
        w6 = (x - xf) * (y - yf) * (zc - z)
        w7 = (x - xf) * (y - yf) * (z - zf)
        w = torch.cat([w0, w1, w2, w3, w4, w5, w6, w7], dim=1)
        w = w.transpose(1, 0).contiguous()
        if scale != 1:
            w /= scale ** 3
        w[idx_query == -1] = 0
        w /= torch.sum(w, dim=0) + 1e-8
    return w


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    """nn/functional/downsample.py (pure torch upstream): floor xyz to multiples of
    stride*tensor_stride, then torch.unique on (b,x,y,z) rows => sorted lexicographically."""
    stride = make_ntuple(stride, ndim=3)
    kernel_size = make_ntuple(kernel_size, ndim=3)
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    assert all(stride[k] in [1, kernel_size[k]] for k in range(3)), \
        'only the stride in {1, kernel_size} branch is on the LiDAL path'
    ss = torch.tensor([stride[k] * tensor_stride[k] for k in range(3)],
                      dtype=torch.int).unsqueeze(0)
    coords = coords.clone()
    coords[:, :3] = torch.div(coords[:, :3], ss, rounding_mode='floor') * ss
    coords = coords[:, [3, 0, 1, 2]]
    coords = torch.unique(coords, dim=0)
    coords = coords[:, [1, 2, 3, 0]]
    return coords.contiguous()


def build_kmap(coords, in_stride, kernel_size, stride):
    """The kernel-map construction inside nn/functional/conv.py::conv3d (cache miss branch).
    Returns (nbmaps i64 [M,2] = (in_idx, out_idx) grouped by k and ordered by out_idx,
             nbsizes i64 [K], (n_in, n_out), out_coords, results i64 [K, n_out])."""
    offsets = get_kernel_offsets(kernel_size, stride=in_stride)
    references = sphash(coords)
    out_coords = coords
    if any(s > 1 for s in stride):
        out_coords = spdownsample(coords, stride, kernel_size, in_stride)
    queries = sphash(out_coords, offsets)
    results = sphashquery(queries, references)
    nbsizes = torch.sum(results != -1, dim=1)
    nbmaps = torch.nonzero(results != -1)
    nbmaps[:, 0] = results.view(-1)[nbmaps[:, 0] * results.size(1) + nbmaps[:, 1]]
    return nbmaps, nbsizes, (coords.shape[0], out_coords.shape[0]), out_coords, results


def _conv_apply(feats, weight, nbmaps, nbsizes, sizes, transposed):
    """nn/functional/conv.py::ConvolutionFunction.forward, CPU branch: per-offset
    index_select -> mm -> index_add.  Written with differentiable torch ops so that autograd
    yields gin[in_map_k] += gout[out_map_k] @ W[k]^T and gW[k] = in[in_map_k]^T @ gout[out_map_k]
    (the CUDA backward upstream; upstream has no CPU backward)."""
    n_out = sizes[0] if transposed else sizes[1]
    output = torch.zeros(n_out, weight.size(-1), dtype=feats.dtype)
    cur = 0
    for k in range(weight.shape[0]):
        n = int(nbsizes[k])
        in_map = nbmaps[cur:cur + n, 0].long()
        out_map = nbmaps[cur:cur + n, 1].long()
        cur += n
        if n == 0:
            continue
        if transposed:
            in_map, out_map = out_map, in_map
        output = output.index_add(0, out_map, torch.mm(feats[in_map], weight[k]))
    return output


def conv3d(input, weight, kernel_size, bias=None, stride=1, dilation=1, transposed=False):
    """nn/functional/conv.py::conv3d.  Reached from every spnn.Conv3d.forward
    (network/utils.py:110-114,129-133,147-155,163-164; network/spvcnn.py:22,24)."""
    feats, coords = input.feats, input.coords
    kernel_size = make_ntuple(kernel_size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)

    if kernel_size == (1, 1, 1) and stride == (1, 1, 1) and dilation == (1, 1, 1):
        feats = feats.matmul(weight)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=coords, feats=feats, stride=input.stride)
    elif not transposed:
        key = (input.stride, kernel_size, stride, dilation)
        kmap = input.kmaps.get(key)
        if kmap is None:
            assert dilation == (1, 1, 1)
            nbmaps, nbsizes, sizes, out_coords, _ = build_kmap(coords, input.stride,
                                                                kernel_size, stride)
            kmap = [nbmaps, nbsizes, sizes, out_coords]
            input.kmaps[key] = kmap
        feats = _conv_apply(feats, weight, kmap[0], kmap[1], kmap[2], transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=kmap[3] if any(s > 1 for s in stride) else coords,
                              feats=feats,
                              stride=tuple(input.stride[k] * stride[k] for k in range(3)))
    else:
        tensor_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        kmap = input.kmaps[(tensor_stride, kernel_size, stride, dilation)]
        feats = _conv_apply(feats, weight, kmap[0], kmap[1], kmap[2], transposed)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(coords=input.cmaps[tensor_stride], feats=feats,
                              stride=tensor_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output
