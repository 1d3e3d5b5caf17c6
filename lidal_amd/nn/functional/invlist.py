"""Per-voxel contributor lists (the transpose of a point->voxel index), built once per index
tensor and cached ON that tensor, so voxelize forward and devoxelize backward run as ordered
per-voxel gathers instead of float atomics: faster on MI355X and bitwise reproducible."""
import torch

from ... import backend as B

__all__ = ['inverse_lists', 'segment_workspace']


def inverse_lists(idx, m, weights=None):
    """idx: i32 tensor [N] or [N,8]; returns (order i32 [idx.numel()], seg_ptr i64 [m+1]).
    Cached as attributes of `idx` (the glue code re-uses one index tensor per stride)."""
    # the list depends on the index CONTENTS and, for devoxelize backward, on which weights are zero:
    # key on version counter + storage of both (an in-place edit or a re-used tensor rebuilds it)
    key = (m, idx._version, idx.data_ptr(), idx.numel()) + \
          (() if weights is None else (weights._version, weights.data_ptr()))
    cached = getattr(idx, '_lidal_invlist', None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    B.require_gpu(idx)
    flat = idx.contiguous().view(-1)
    assert flat.dtype == torch.int
    n = flat.numel()
    order = B.empty(max(n, 1), torch.int, idx.device)
    seg_ptr = B.empty(m + 1, torch.int64, idx.device)
    ws_bytes = B.lib().lidal_invlist_workspace_bytes(n)
    ws = B.workspace(ws_bytes, idx.device)
    w = None if weights is None else weights.contiguous().view(-1)
    B.check(B.lib().lidal_invlist_build(B.ptr(flat), B.ptr(w), n, m, B.ptr(order), B.ptr(seg_ptr),
                                        B.ptr(ws), ws_bytes, B.stream()), 'invlist_build')
    idx._lidal_invlist = (key, order, seg_ptr)
    return order, seg_ptr


def segment_workspace(n_entries, m, c, device):
    """(buffer or None, bytes) for the split-list form of the ordered segment sums."""
    nbytes = B.lib().lidal_segment_workspace_bytes(n_entries, m, c)
    if nbytes == 0:
        return None, 0
    return B.workspace(nbytes, device), nbytes        # (one library call's scratch)
