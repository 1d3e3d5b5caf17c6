"""ctypes binding of liblidal_amd.so (include/lidal_amd.h) for torch device tensors.

This is the only place that touches the shared library.  There is NO CPU fallback: if the
library is missing, or a tensor is not on the GPU, the call raises.  (The CPU restatement of the
path lives in oracle/ and is test infrastructure only.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LIDAL_AMD_LIB') or os.path.join(_HERE, 'liblidal_amd.so')    # override: A/B builds

F32, BF16 = 0, 1
_lib = None
# calls that reached the HIP library, per entry point family (tests assert the product modules took
# the HIP path: spnn.Linear / BatchNorm1d can fall through to torch for configurations the kernels
# do not cover, and a silent dispatch regression would otherwise still pass the numerics tests)
HITS = {}


def hit(name):
    HITS[name] = HITS.get(name, 0) + 1

_vp, _i32, _i64, _f32, _f64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                               ctypes.c_double)

# name -> (restype, argtypes): every symbol include/lidal_amd.h declares
SIGNATURES = {
    'lidal_last_error': (ctypes.c_char_p, []),
    'lidal_version': (_i32, []),
    'lidal_floor_coords': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'lidal_revoxelize_coords': (_i32, [_vp, _i64, _f32, _f32, _vp, _vp, _vp]),
    'lidal_hash': (_i32, [_vp, _i64, _vp, _vp]),
    'lidal_kernel_hash': (_i32, [_vp, _i64, _vp, _i32, _vp, _vp]),
    'lidal_hash_table_bytes': (_i64, [_i64]),
    'lidal_hash_table_build_coords': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp]),
    'lidal_hash_table_build': (_i32, [_vp, _i64, _vp, _i64, _vp]),
    'lidal_hash_table_query': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'lidal_unique_workspace_bytes': (_i64, [_i64]),
    'lidal_unique_sorted_i64': (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    'lidal_downsample_workspace_bytes': (_i64, [_i64]),
    'lidal_downsample': (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    'lidal_downsample_pyramid_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_downsample_pyramid': (_i32, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    'lidal_voxelize_points_workspace_bytes': (_i64, [_i64]),
    'lidal_voxelize_points': (_i32, [_vp, _vp, _i64, _vp, _vp, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _i64, _vp]),
    'lidal_kmap_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_kmap_build': (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp,
                                _i64, _vp]),
    'lidal_kmap_build_batch_workspace_bytes': (_i64, [_vp, _vp, _i32]),
    'lidal_kmap_build_batch': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp]),
    'lidal_kmap_invert': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp]),
    'lidal_kmap_from_rules': (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp]),
    'lidal_count': (_i32, [_vp, _i64, _vp, _i64, _vp]),
    'lidal_voxelize_fwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    'lidal_voxelize_fwd_1to1': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    'lidal_voxelize_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    'lidal_devoxelize_fwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    'lidal_devoxelize_bwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    'lidal_invlist_workspace_bytes': (_i64, [_i64]),
    'lidal_invlist_build': (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    'lidal_segment_workspace_bytes': (_i64, [_i64, _i64, _i32]),
    'lidal_voxelize_fwd_sorted': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp]),
    'lidal_devoxelize_bwd_sorted': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp]),
    'lidal_devoxelize_bwd_cells': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    'lidal_devoxelize_bwd_cells_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_ti_weights': (_i32, [_vp, _i32, _vp, _i64, _f32, _vp, _vp, _vp]),
    'lidal_sort_pairs_workspace_bytes': (_i64, [_i64]),
    'lidal_sort_pairs': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    'lidal_sort_pairs_u64': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    'lidal_kmap_order_workspace_bytes': (_i64, [_i64]),
    'lidal_kmap_order': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'lidal_kmap_order_batch': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'lidal_conv_stats_tile_rows': (_i32, []),
    'lidal_conv_weight_image_bytes': (_i64, [_i32, _i32, _i32, _i32, _i64]),
    'lidal_conv_weight_image_tiling': (_i32, [_i32, _i32, _i32, _i64]),
    'lidal_conv_weight_image': (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i64, _vp]),
    'lidal_conv_weight_image_pair': (_i32, [_vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    'lidal_conv_weight_image_job_bytes': (_i32, []),
    'lidal_conv_weight_image_job': (_i64, [_vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i64]),
    'lidal_conv_weight_image_batch': (_i32, [_vp, _i32, _i64, _i32, _i32, _vp]),
    'lidal_conv_apply_image': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp,
                                      _vp, _i32, _vp, _vp, _vp]),
    'lidal_conv_dgrad_bn_sums': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp,
                                        _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    'lidal_conv_apply_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_conv_apply_image_ws': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp,
                                         _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    'lidal_conv_dgrad_bn_sums_ws': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp,
                                           _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    'lidal_conv_wgrad_slabs': (_i64, [_i64, _i64, _i32, _i32, _i32, _i32]),
    'lidal_conv_wgrad': (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    'lidal_conv_wgrad_streams': (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    'lidal_conv_wgrad_streams_serves': (_i32, [_i64, _i64, _i32, _i32, _i32]),
    'lidal_wgrad_streams_workgroups': (_i32, []),
    'lidal_wgrad_streams_rules': (_i64, [_i64, _i32, _i32]),
    'lidal_wgrad_streams_desc_words': (_i64, [_i32, _i32]),
    'lidal_wgrad_streams_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_wgrad_streams_build': (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp]),
    'lidal_bn_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_bn_train_fwd': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                  _vp, _vp, _i64, _vp]),
    'lidal_bn_train_fwd_tiles': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _i32, _vp, _vp,
                                        _vp, _vp, _vp, _i64, _vp]),
    'lidal_bn_eval_fwd': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp]),
    'lidal_bn_bwd': (_i32, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                            _i64, _vp]),
    'lidal_bn_bwd_tiles': (_i32, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _i64, _vp]),
    'lidal_add_relu_bwd_bn_sums': (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'lidal_add_relu_bwd_bn_tile_sums': (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'lidal_bn_tail_parts': (_i64, [_i64, _i32, _i32]),
    'lidal_bn_bwd_from_sums': (_i32, [_vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _i64, _vp]),
    'lidal_bn_set_fused': (_i32, [_i32]),
    'lidal_bn_set_slab_sums': (_i32, [_i32]),
    'lidal_bn_check_device': (_i32, []),
    'lidal_bn_fold': (_i32, [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp]),
    'lidal_colsum': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _i64, _vp]),
    'lidal_add_relu_fwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    'lidal_add_relu_bwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    'lidal_ce_workspace_bytes': (_i64, [_i64]),
    'lidal_ce_fwd': (_i32, [_vp, _i32, _vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp]),
    'lidal_ce_bwd': (_i32, [_vp, _i32, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    'lidal_view_mean_softmax': (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    'lidal_confusion_accumulate': (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    'lidal_register_points': (_i32, [_vp, _i64, _vp, _vp, _vp]),
    'lidal_nn_grid_bytes': (_i64, [_i64]),
    'lidal_nn_grid_workspace_bytes': (_i64, [_i64]),
    'lidal_nn_grid_build': (_i32, [_vp, _i64, _f64, _vp, _i64, _vp, _i64, _vp]),
    'lidal_interframe_workspace_bytes': (_i64, [_i64, _i32]),
    'lidal_interframe_score': (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _f64, _vp,
                                      _vp, _vp, _vp, _i64, _vp]),
    'lidal_interframe_score_ordered': (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _f64, _vp,
                                              _vp, _vp, _vp, _i64, _vp, _vp]),
    'lidal_supervoxel_reduce': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'lidal_copy2d': (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _vp]),
    'lidal_add2d': (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    'lidal_transpose_f32': (_i32, [_vp, _i64, _vp, _i32, _i32, _vp]),
    'lidal_cast_rows_bf16': (_i32, [_vp, _i32, _vp, _i32, _i64, _vp]),
    'lidal_plan_op_args': (_i32, [_i32]),
    'lidal_debug_read': (_i32, [_vp, _vp, _i64]),
    'lidal_plan_run': (_i32, [_vp, _i64, _i64, _vp, _vp]),
    'lidal_plan_run_streams': (_i32, [_vp, _i64, _i64, _vp, _i32]),
}


class _TimedLib:
    """Measurement hook (bench.py): every library call bracketed by events on the launch stream and
    reported to `sink(name, args, e0, e1)`.  Off unless set_call_timer() installs a sink."""

    def __init__(self, handle, sink):
        self._handle, self._sink = handle, sink

    def __getattr__(self, name):
        fn = getattr(self._handle, name)
        if fn.restype is not _i32 or name.endswith(('_bytes', '_tiling', '_rows', '_workgroups', '_serves')) or name == 'lidal_version':
            return fn                               # size queries: no kernel behind them
        sink = self._sink

        def timed(*args):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            sink(name, args, e0, e1)
            return rc
        return timed


_timer = None


def set_call_timer(sink):
    """sink(name, args, start_event, end_event) per library call, or None to switch timing off."""
    global _timer
    _timer = None if sink is None else _TimedLib(lib_handle(), sink)


def lib_handle():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'lidal_amd: %s is missing -- build it with `python -m lidal_amd.build` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def lib():
    return _timer if _timer is not None else lib_handle()


def check(rc, what):
    HITS[what] = HITS.get(what, 0) + 1
    if rc != 0:
        raise RuntimeError('lidal_amd.%s failed (%d): %s' %
                           (what, rc, lib().lidal_last_error().decode()))


def bind_cpus_near(device_index=0):
    """Restrict EVERY thread of this process (the caller's, autograd's device thread if it exists already, the ones
    started later inherit) to the CPUs of the NUMA node the GPU hangs off (`/sys/bus/pci/devices/<bdf>/local_cpulist`) --
    what `numactl --cpunodebind` does for a launcher with one process per GPU.  The single-scan step is host-bound: measured on a 2-socket host (scripts/exp/
    host_mode_probe.py) 6.54 +- 0.02 ms bound against 6.7-7.9 ms wherever the scheduler put the process.  Returns the
    CPU set, or None when the topology cannot be read (nothing changes then)."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        text = open('/sys/bus/pci/devices/%s/local_cpulist' % bdf).read().strip()
        cpus = set()
        for part in text.split(','):
            if '-' in part:
                a, b = part.split('-')
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        for tid in os.listdir('/proc/self/task'):           # threads that exist already (autograd's device thread runs the
            try:                                            # backward plan's host work)
                os.sched_setaffinity(int(tid), cpus)
            except (OSError, ValueError):
                pass
        return cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def require_gpu(*tensors):
    """Every operand must be a GPU tensor ON THE CURRENT DEVICE: kernels are launched on the current
    device's stream (stream() below), so a tensor of another GPU would be a foreign pointer there
    (a memory fault, not a Python error).  One process per GPU is the deployment; inside one
    process use `with torch.cuda.device(t.device):` around calls on a non-current device."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                'lidal_amd operators run on the GPU only (got a %s tensor); the CPU restatement '
                'is oracle/ and is test infrastructure, not a fallback' % t.device)
        if cur is None:
            cur = torch._C._cuda_getDevice()
        if t.device.index != cur:
            raise RuntimeError('lidal_amd: tensor on %s but the current device is cuda:%d -- kernels '
                               'launch on the current device (torch.cuda.set_device / '
                               'torch.cuda.device(...))' % (t.device, cur))


# ---- memory of the coordinate work ------------------------------------------------------------------------
# The tables a forward pass derives from its coordinates (kernel maps, row orders, hash tables, contributor lists,
# the voxeliser's outputs) have sizes that follow the batch's voxel counts, and under the reference's per-iteration
# augmentation (dataset/sk_dataset.py:143-171) no two batches have the same counts: as ~150 torch.empty calls of
# never-repeating sizes they fragmented the caching allocator's pools (the reserved bytes of a never-repeating input
# stream kept growing, bench.py variants.fresh_stream).  So the builders ask `empty()` / `workspace()`:
#   * `empty` carves buffers of 1 MiB or more out of the blocks of the BlockArena that is current (network/geometry.py
#     gives every Geometry its own: ONE block size, so the allocator re-uses the blocks exactly, whatever the counts;
#     the tables of a geometry die together, and so do their blocks); without a current arena it rounds the size up to
#     1/8 .. 1/16 steps of its power of two;
#   * `workspace` is the scratch of ONE library call (sort buffers, scan partials): a persistent buffer per stream,
#     grown when a call asks for more (calls of a stream run one after the other).
_QUANT_MIN = 1 << 20
_ARENA = [None]
_WORKSPACE = {}         # (device index, stream) -> uint8 tensor


class BlockArena:
    BLOCK = int(os.environ.get('LIDAL_TABLE_BLOCK_MB', '512')) << 20

    def __init__(self):
        self.blocks = []
        self.cur = None
        self.off = 0

    def take(self, nbytes, device):
        nbytes = (nbytes + 255) & -256
        if nbytes > (self.BLOCK >> 1):              # a block of its own, in 64 MiB steps
            t = torch.empty((nbytes + (64 << 20) - 1) & -(64 << 20), dtype=torch.uint8, device=device)
            self.blocks.append(t)
            return t[:nbytes]
        if self.cur is None or self.off + nbytes > self.BLOCK or self.cur.device != device:
            self.cur = torch.empty(self.BLOCK, dtype=torch.uint8, device=device)
            self.blocks.append(self.cur)
            self.off = 0
        v = self.cur[self.off:self.off + nbytes]
        self.off += nbytes
        return v


class use_arena:
    """with use_arena(arena): the large buffers of the coordinate builders come out of `arena`'s blocks."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        self.saved = _ARENA[0]
        _ARENA[0] = self.arena
        return self.arena

    def __exit__(self, *exc):
        _ARENA[0] = self.saved
        return False


def empty(shape, dtype, device):
    if isinstance(shape, int):
        shape = (shape,)
    n = dtype.itemsize
    for d in shape:
        n *= d
    if n < _QUANT_MIN:
        return torch.empty(shape, dtype=dtype, device=device)
    arena = _ARENA[0]
    if arena is not None:
        return arena.take(n, torch.device(device))[:n].view(dtype).view(shape)
    q = 1 << (n.bit_length() - 4)
    return torch.empty((n + q - 1) & -q, dtype=torch.uint8, device=device)[:n].view(dtype).view(shape)


def workspace(nbytes, device):
    """uint8 scratch of at least `nbytes` for ONE library call on the current stream of `device` (valid until the
    next workspace() request on that stream is used)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    t = _WORKSPACE.get(key)
    if t is None or t.numel() < nbytes:
        size = max(16 << 20, (int(nbytes * 1.25) + (1 << 20)) & -(1 << 20))
        with torch.cuda.device(idx):
            t = _WORKSPACE[key] = torch.empty(size, dtype=torch.uint8, device=device)
    return t


def ptr(t):
    """Device address of a tensor (None -> NULL) as a plain int: ctypes converts it for the c_void_p
    argtypes itself, and a step makes ~1300 of these (a c_void_p object each was 0.3 ms per step)."""
    return None if t is None else t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device.  Goes through the C
    accessor directly: torch.cuda.current_stream() builds a Python Stream object (~6 us), and the
    path makes ~230 library calls per training step."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())        # (a plain int: see ptr)


# ---- when may cached weight operands be re-used? -----------------------------------------------
# LDS images of weights, folded BatchNorm maps etc. are cached per parameter.  A parameter's version
# counter moves with `copy_` / `load_state_dict` / the foreach optimizers -- but NOT with
# torch.optim.Adam(fused=True) (measured: five fused steps left `_version` where it was, and a bank keyed
# on it alone trained on the first step's images), nor with writes through `.data`.  So the key also
# carries an EPOCH that moves at the end of every backward pass that went through this library's
# operators: parameters only change between a backward pass and the next forward.
WEIGHT_EPOCH = [0]
_epoch_queued = [False]
_epoch_task = [-1]          # id of the autograd graph task whose end-of-backward callback is queued


def _graph_task():
    return torch._C._current_graph_task_id()        # -1 outside a backward pass


def _bump_epoch():
    WEIGHT_EPOCH[0] += 1
    _epoch_queued[0] = False


def _settle():
    """A backward pass that RAISED never ran its final callbacks (the engine drops them): the latches
    would stay set for the rest of the process -- the epoch frozen (stale weight images under
    Adam(fused=True)) and the side-stream events never waited for.  Called where a stale latch can be
    seen: from the next backward pass (a different graph task) and from the next forward (no graph task)."""
    if _epoch_queued[0] and _graph_task() != _epoch_task[0]:
        _bump_epoch()
    if _join_queued[0] and _graph_task() != _join_task[0]:
        join_side_streams()


def note_backward():
    """Called from the operators' backward functions: the epoch moves when this backward pass ends."""
    _settle()
    if not _epoch_queued[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_bump_epoch)
            _epoch_queued[0] = True
            _epoch_task[0] = _graph_task()
        except RuntimeError:            # not inside a backward pass (a backward function called by hand)
            _bump_epoch()


def weights_key(t):
    """What a cached operand of parameter / buffer `t` is valid for."""
    if _epoch_queued[0] or _join_queued[0]:
        _settle()
    return (WEIGHT_EPOCH[0], t._version, t.data_ptr())


# ---- second stream for the weight gradients --------------------------------------------------
# In the backward pass of a convolution the weight gradient and the data gradient both depend only
# on grad_out; nothing downstream needs the weight gradient before the optimizer (or DDP's bucket
# hook).  The weight gradient runs one 4-wave workgroup per CU on the large levels and a handful of
# workgroups on the small ones, so it is launched on a second stream beside the data gradient and the
# BatchNorm backward that follows it.  Measured on the bench batch (5 scans): the f32 step goes from
# 69.4 to 56.2 ms (its MFMA-bound kernels leave the memory system to the other stream); the bf16 step
# does not move (21.5 -> 21.8 ms: both kernels are bound by the same cache-line traffic) and a single
# scan gets slower (12.7 -> 15.7 ms, host-bound: the stream switches cost Python time).  So: f32 only
# by default; LIDAL_WGRAD_STREAM=0 / 1 forces it off / on for every dtype.
_side_streams = {}
_pending = []
_join_queued = [False]
_join_task = [-1]
_seen_weights = set()        # id(weight) of the weight gradients launched on the side stream, this backward pass
_OVERLAP = os.environ.get('LIDAL_WGRAD_STREAM', 'auto')


# gradient fan-in fused into the producing kernels (residual-block input, point features); 0 = off
FORK = int(os.environ.get('LIDAL_FORK', '15'))      # bit 0: residual blocks, bit 1: point features, bit 2: point-branch
                                                    # sum in BatchNorm, bit 3: relu(bn + shortcut) in BatchNorm


_OVERLAP_ROWS = int(os.environ.get('LIDAL_WGRAD_STREAM_ROWS', '0'))

# The streamed weight gradient (round 6; csrc/wgrad_streams.hip): the rule lists of a stride-1 map re-ordered into one
# stream per workgroup, so that the rules of a row meet in one XCD's L2 (nn/functional/conv.py KernelMap.streams).  Taken
# on levels with at least this many rows; 0 = never, THE DEFAULT.  Measured (profiles/README.md, round 6): alone on the
# GPU a launch on the 397 k / 226 k-row levels of a 5-scan batch takes 81 / 68 us instead of 130 / 95 (FETCH_SIZE 2 x 164 MB
# instead of 2 x 425, L2 hits 65 % instead of 5 %), the tables cost 0.27 / 0.18 ms per map and step; inside the training
# step -- where the weight gradients run beside the data gradients, whose waves leave a SIMD's registers no room for a
# streamed wave with its two accumulator sets -- the streams' workgroups are no longer resident together, the blocks they
# should share are gone from the L2 when the late ones arrive (177 us per launch against 190), and the step is 0.1 ms SLOWER
# (13.58 -> 13.68 ms) with the tables' launches on the second queue.  LIDAL_WGRAD_STREAMS_ROWS=150000 takes it on the two finest levels.
WGRAD_STREAMS_ROWS = int(os.environ.get('LIDAL_WGRAD_STREAMS_ROWS', '0'))


def overlap_wgrad(dtype, n_rows=0):
    # (round 5: no longer for f32 under 'auto' -- with the weight gradients beside the data gradients one SPVCNN f32 run
    #  in ~80 was not bit-reproducible, profiles/README.md "A training run that was not bit-reproducible"; '1' forces it)
    if _OVERLAP == '1':
        return True
    # experiment: bf16 layers of the coarse levels only (their kernels do not fill the chip)
    return _OVERLAP == 'auto' and 0 < n_rows <= _OVERLAP_ROWS


def side_stream(device, which=1):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if which != 1:
        key = key + (which,)
    if key not in _side_streams:
        # (LIDAL_X_SIDE_PRIORITY=-1: the weight gradients' stream as a high-priority queue -- an experiment, profiles/README.md round 6)
        pr = int(os.environ.get('LIDAL_X_SIDE_PRIORITY', '0')) if which == 1 else 0
        _side_streams[key] = torch.cuda.Stream(device=device, priority=pr)
    return _side_streams[key]


def join_side_streams():
    """Make the current stream wait for every weight gradient launched beside it.  Idempotent."""
    _join_queued[0] = False
    _seen_weights.clear()
    if _pending:
        cur = torch.cuda.current_stream()
        for ev in _pending:
            cur.wait_event(ev)
        del _pending[:]


def _may_defer(weight):
    """May the main stream's wait for this weight gradient be put off to the end of the backward
    pass?  Only if nothing on the main stream can touch the gradient before then: autograd believes
    it was produced on the main stream, so an accumulation into an existing `.grad` (gradient
    accumulation, zero_grad(set_to_none=False)), the sum of two uses of a shared weight, a tensor or
    post-accumulate hook, or DDP's bucket copy (at ANY world size) would read it unsynchronised."""
    import torch.distributed as dist
    if weight is None or weight.grad is not None:
        return False
    if dist.is_available() and dist.is_initialized():
        return False
    if getattr(weight, '_backward_hooks', None) or getattr(weight, '_post_accumulate_grad_hooks', None):
        return False
    if id(weight) in _seen_weights:         # second use of a shared weight in this backward pass
        return False
    return True


def beside(device, inputs, weight=None):
    """`weight`: the parameter whose gradient is produced (decides whether the join may be deferred,
    _may_defer; None = join at once).  Context for work that may run beside the current stream: `with beside(dev, (x, g)) as done:`
    ... launch ...; `done(outputs)`.  The side stream first waits for everything already enqueued on
    the current stream (so call this BEFORE enqueueing the work it should overlap with); the current
    stream waits for the side work at the end of the backward pass -- or, when gradients are reduced
    across ranks by hooks that fire during the backward pass, at once."""
    return _Beside(device, inputs, weight)


class _Beside:
    def __init__(self, device, inputs, weight=None):
        self.weight = weight
        self.main = torch.cuda.current_stream(device)
        self.side = side_stream(device)
        self.inputs = inputs

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self._done

    def _done(self, *outputs):
        for t in self.inputs:
            if t is not None:
                t.record_stream(self.side)
        for t in outputs:
            t.record_stream(self.main)
        self.event = self.side.record_event()

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        if exc[0] is not None:
            return False
        if not _may_defer(self.weight):
            self.deferred = False
            if self.weight is not None and id(self.weight) in _seen_weights:
                join_side_streams()         # the earlier gradient of this shared weight is summed with this one
        else:
            self.deferred = True
            _seen_weights.add(id(self.weight))
            _pending.append(self.event)
            if not _join_queued[0]:
                try:          # inside a backward pass: join when it ends
                    torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
                    _join_queued[0] = True
                    _join_task[0] = _graph_task()
                except RuntimeError:
                    join_side_streams()
        return False

    def finish(self):
        """Call after the overlapping work has been enqueued on the current stream."""
        if not self.deferred:
            self.main.wait_event(self.event)


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError('lidal_amd: unsupported feature dtype %s (float32 / bfloat16 only)' % dt)


# f32 INFERENCE runs its convolutions and dense layers in the split form (include/lidal_amd.h: LIDAL_F32_SPLIT; csrc/
# conv_img.hip conv_split_kernel): f32 features and results, every operand cut exactly into three bf16 pieces, six partial
# products on the bf16 matrix cores, f32 accumulation -- within 3 * 2^-24 per product of the exact-f32 kernel, which the
# training step's f32 parity mode keeps.  LIDAL_F32_SPLIT=0: the exact f32 MFMA everywhere.
F32_SPLIT = 2
SPLIT_F32 = os.environ.get('LIDAL_F32_SPLIT', '1') != '0'
# Round 6: the f32 TRAINING step (the reference trains in fp32, train.py:127-140) runs the forward products and the data
# gradients of its sparse convolutions in the split form too, wherever BOTH channel counts are whole 32-channel slices
# (one rule per layer: its forward reduces over ci, its data gradient over co) -- everything but the 4-channel stem
# convolution; the weight gradients and the dense layers (1x1x1 convolutions, nn.Linear) keep the exact f32 MFMA.  As
# close to the f64 oracle as the exact kernels (tests/test_benchsize_gpu.py, same calibrated bars).
# LIDAL_F32_SPLIT_TRAIN=0: the exact f32 MFMA everywhere in training (the association the round-1..5 numbers were taken with).
SPLIT_F32_TRAIN = os.environ.get('LIDAL_F32_SPLIT_TRAIN', '1') != '0'


def conv_code(dt, n_red, inference, n_col=None):
    """dtype code of a convolution / dense product over `n_red` input channels on features of dtype `dt`.
    n_col (training, sparse convolutions only): the layer's other channel count."""
    if dt == torch.float32 and SPLIT_F32 and n_red > 0 and n_red % 32 == 0:
        if inference:
            return F32_SPLIT
        if (SPLIT_F32_TRAIN and n_col is not None and n_col > 0 and n_col % 32 == 0
                and os.environ.get('LIDAL_X_SPLIT_APPLY') != '0'):
            return F32_SPLIT
    return dtype_code(dt)


def wgrad_code(dt, ca, cb):
    """dtype code of a weight gradient a^T b (a [n, ca], b [n, cb] of dtype `dt`): f32 operands take the split form
    (lidal_conv_wgrad with LIDAL_F32_SPLIT: the operands cut into three bf16 pieces inside the call, six bf16 MFMAs per
    product -- round 6) wherever both channel counts are whole 16-byte segments; LIDAL_F32_SPLIT_TRAIN=0: exact f32."""
    if dt == torch.float32 and SPLIT_F32 and SPLIT_F32_TRAIN and ca > 0 and cb > 0 and ca % 8 == 0 and cb % 8 == 0:
        return F32_SPLIT
    return dtype_code(dt)


def wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class NoGradCtx:
    """Stand-in for an autograd ctx: lets a Function's forward body run directly (no autograd node,
    nothing kept alive) when no input wants a gradient, i.e. at inference."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *tensors):
        pass


def compute_dtype(x):
    """Operand dtype of the MFMA kernels for input `x`: bf16 under bf16 autocast, the input's own
    dtype if it is f32 / bf16, f32 otherwise (fp16 autocast is not a mode of this backend: it
    computes in f32 rather than silently in another 16-bit format)."""
    if torch.is_autocast_enabled():
        return torch.bfloat16 if torch.get_autocast_dtype('cuda') == torch.bfloat16 else torch.float32
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32


_STATS_ROWS = [0]


def stats_tile_rows():
    """Rows per tile of the BatchNorm statistics the convolution kernels leave (asked of the library once)."""
    if not _STATS_ROWS[0]:
        _STATS_ROWS[0] = int(lib_handle().lidal_conv_stats_tile_rows())
    return _STATS_ROWS[0]
