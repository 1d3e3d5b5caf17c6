"""F.spdownsample (torchsparse/nn/functional/downsample.py) and the sorted unique of hashes that
network/utils.py:18 takes with torch.unique.  Both produce SORTED output (that order defines the
voxel row order of every coarser level), so they are radix-sort + ordered compaction on device."""
import torch

from ... import backend as B
from ...utils import make_ntuple

__all__ = ['spdownsample', 'unique_sorted', 'downsample_pyramid']


def unique_sorted(keys):
    """Sorted unique of an i64 key vector (== torch.unique(keys) for non-negative keys)."""
    B.require_gpu(keys)
    keys = keys.contiguous().view(-1)
    assert keys.dtype == torch.int64
    n = keys.numel()
    out = B.empty(n, torch.int64, keys.device)
    n_out = torch.empty(1, dtype=torch.int64, device=keys.device)
    ws_bytes = B.lib().lidal_unique_workspace_bytes(n)
    ws = B.workspace(ws_bytes, keys.device)
    B.check(B.lib().lidal_unique_sorted_i64(B.ptr(keys), n, B.ptr(out), B.ptr(n_out), B.ptr(ws),
                                            ws_bytes, B.stream()), 'unique_sorted_i64')
    return out[:int(n_out.item())]


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    stride = make_ntuple(stride, ndim=3)
    kernel_size = make_ntuple(kernel_size, ndim=3)
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    if not all(stride[k] in [1, kernel_size[k]] for k in range(3)):
        raise NotImplementedError('spdownsample: only stride in {1, kernel_size} is on the LiDAL '
                                  'path (network/spvcnn.py:28,34,40,46 use 2/2)')
    B.require_gpu(coords)
    assert coords.dtype == torch.int and coords.shape[1] == 4
    coords = coords.contiguous()
    n = coords.shape[0]
    ss = [stride[k] * tensor_stride[k] for k in range(3)]
    out = B.empty((n, 4), torch.int, coords.device)
    n_out = torch.empty(1, dtype=torch.int64, device=coords.device)
    ws_bytes = B.lib().lidal_downsample_workspace_bytes(n)
    ws = B.workspace(ws_bytes, coords.device)
    B.check(B.lib().lidal_downsample(B.ptr(coords), n, ss[0], ss[1], ss[2], B.ptr(out),
                                     B.ptr(n_out), B.ptr(ws), ws_bytes, B.stream()), 'downsample')
    return out[:int(n_out.item())]


def downsample_pyramid(coords, levels, tensor_stride=1):
    """[spdownsample(spdownsample(... coords ...))] for `levels` chained stride-2 / kernel-2 downsamplings
    (the encoder of network/spvcnn.py:28-46), all from one sort of the input voxels and with one host
    round trip for all row counts: a list of `levels` coordinate tensors (views of one buffer), the l-th
    at tensor stride 2^(l+1) * tensor_stride.  Same rows, same order as the chain.  Raises ValueError for coordinates
    outside 0 <= x, y, z < 65536 / 0 <= batch < 8192 (checked on the device, reported with the row counts)."""
    tensor_stride = make_ntuple(tensor_stride, ndim=3)
    B.require_gpu(coords)
    assert coords.dtype == torch.int and coords.shape[1] == 4 and 1 <= levels <= 4
    coords = coords.contiguous()
    n = coords.shape[0]
    out = B.empty((max(n * levels, 1), 4), torch.int, coords.device)
    starts = torch.empty(levels + 1, dtype=torch.int64, device=coords.device)
    ws_bytes = B.lib().lidal_downsample_pyramid_workspace_bytes(n, levels)
    ws = B.workspace(ws_bytes, coords.device)
    B.check(B.lib().lidal_downsample_pyramid(B.ptr(coords), n, tensor_stride[0], tensor_stride[1], tensor_stride[2],
                                             levels, B.ptr(out), B.ptr(starts), B.ptr(ws), ws_bytes, B.stream()),
            'downsample')
    st = starts.tolist()
    if st[-1] < 0:          # the packed sort key has 16 bits per coordinate and 13 for the batch index
        raise ValueError('lidal_amd: downsample_pyramid needs 0 <= x, y, z < 65536 and 0 <= batch < 8192 (the '
                         'reference feeds voxel coordinates shifted into [0, full_scale), dataset/sk_dataset.py:156-161)')
    return [out[st[l]:st[l + 1]] for l in range(levels)]
