"""F.conv3d (torchsparse/nn/functional/conv.py v1.4.0) on the HIP backend.

Reached from every spnn.Conv3d.forward (network/utils.py:110-114,129-133,147-155,163-164;
network/spvcnn.py:22,24).  Three cases, as upstream:
  * 1x1x1 stride 1          -> feats @ weight (dense library GEMM, as upstream)
  * regular / strided       -> kernel map looked up in input.kmaps, built on a miss
  * transposed              -> re-uses the map of the matching strided conv with roles swapped

KernelMap holds what torchsparse keeps ([nbmaps, nbsizes, (n_in, n_out)], same order and
values) plus what the output-stationary kernel consumes:
  nbr_out [K, n_out]  input row feeding output row j through offset k (-1: none)
  nbr_in  [K, n_in]   output row fed by input row i through offset k (built lazily)
  order_out/order_in  RowOrder: the tables with rows sorted by occupancy pattern (built lazily)
"""
import os

import torch
from torch.autograd import Function

from ... import backend as B
from ...tensor import SparseTensor
from ...utils import make_ntuple
from ..utils import get_kernel_offsets
from .downsample import spdownsample
from .hash import sphash
from .query import coords_table

__all__ = ['conv3d', 'KernelMap', 'RowOrder', 'build_kernel_map', 'prefetch_kernel_maps', 'conv_backward']


class RowOrder:
    """A neighbour table re-ordered by occupancy pattern (lidal_kmap_order): `perm` maps sorted
    position -> row, `table` = nbr[:, perm].  Built once per table, shared by every conv using it."""

    def __init__(self, nbr, build=True):
        k, n = nbr.shape
        dev = nbr.device
        self.n_rows = n
        self.perm = B.empty(max(n, 1), torch.int, dev)
        self.table = B.empty((k, n), torch.int, dev)
        self.tile_masks = B.empty(max(1, -(-n // 128)), torch.int32, dev)
        if build:
            ws_bytes = B.lib().lidal_kmap_order_workspace_bytes(n)
            ws = B.workspace(ws_bytes, dev)
            B.check(B.lib().lidal_kmap_order(B.ptr(nbr), n, k, B.ptr(self.perm), B.ptr(self.table),
                                             B.ptr(self.tile_masks), B.ptr(ws), ws_bytes, B.stream()),
                    'kmap_order')

    @staticmethod
    def build_many(tables):
        """RowOrders of several neighbour tables: the tables of one kernel volume go through ONE key
        array and one sort (lidal_kmap_order_batch, <= 16 tables per call) -- a U-Net's 13 row orders
        per step are two calls instead of 13 chains of small launches.  Same results as RowOrder(t)."""
        import ctypes
        out = [None] * len(tables)
        by_k = {}
        for i, t in enumerate(tables):
            by_k.setdefault((t.shape[0], str(t.device)), []).append(i)
        for (k, _), ids in by_k.items():
            for c0 in range(0, len(ids), 16):
                chunk = ids[c0:c0 + 16]
                if len(chunk) == 1:
                    out[chunk[0]] = RowOrder(tables[chunk[0]])
                    continue
                orders = [RowOrder(tables[i], build=False) for i in chunk]
                n = len(chunk)
                total = sum(o.n_rows for o in orders)
                dev = tables[chunk[0]].device
                ws_bytes = B.lib().lidal_kmap_order_workspace_bytes(total)
                ws = B.workspace(ws_bytes, dev)
                vp, i64 = ctypes.c_void_p * n, ctypes.c_int64 * n
                B.check(B.lib().lidal_kmap_order_batch(
                    vp(*[tables[i].data_ptr() for i in chunk]), i64(*[o.n_rows for o in orders]), n, k,
                    vp(*[o.perm.data_ptr() for o in orders]), vp(*[o.table.data_ptr() for o in orders]),
                    vp(*[o.tile_masks.data_ptr() for o in orders]), B.ptr(ws), ws_bytes, B.stream()),
                    'kmap_order')
                for i, o in zip(chunk, orders):
                    out[i] = o
        return out


class KernelMap:
    """nbr_out is built eagerly; the torchsparse-order rule lists (nbmaps / nbsizes / koff), which
    only the weight gradient and the torchsparse list view read, are derived from it on first use
    (so inference never builds or allocates them: K * n_out * 8 bytes at the fine levels)."""

    def __init__(self, nbr_out, sizes, volume, symmetric):
        self.nbr_out = nbr_out               # i32 [K, n_out]
        self.sizes = sizes                   # (n_in, n_out)
        self.volume = volume
        self.symmetric = symmetric           # odd kernel, stride 1: nbr_in[k] == nbr_out[K-1-k]
        self._rules = None                   # (nbmaps_cap i32 [K*n_out, 2], nbsizes i32 [K], koff i64 [K+1])
        self._nbr_in = None
        self._total = None
        self._order_out = None
        self._order_in = None
        self._owner = None                   # weakref to the network.geometry.Geometry that built the map ahead, if any
        self._stream_key = None              # (i32 [key_k, n] table, key_k, key_range): the rows' spatial key, if not their index
        self._streams = None                 # (spairs, sdesc, n_wg): the rule lists as one stream per workgroup

    def check_alive(self):
        """A map built ahead on a second stream lives until its prefetcher's fence (network/geometry.py): a backward pass
        queued later than the contract allows would read memory that may already serve another build -- refuse it."""
        g = self._owner() if self._owner is not None else None
        if g is not None and not g.alive():
            raise RuntimeError('lidal_amd: this backward pass reads the coordinate tables of a geometry that is stale -- two '
                               'newer ones have been submitted to its prefetcher since (queue forward AND backward of a '
                               'batch before the second submit() after its own: network/geometry.py GeometryPrefetcher)')

    def _build_rules(self):
        if self._rules is None:
            k, n_out = self.nbr_out.shape
            dev = self.nbr_out.device
            nbmaps = B.empty((k * n_out, 2), torch.int, dev)
            nbsizes = torch.empty(k, dtype=torch.int, device=dev)
            koff = torch.empty(k + 1, dtype=torch.int64, device=dev)
            ws_bytes = B.lib().lidal_kmap_workspace_bytes(n_out, k)
            ws = B.workspace(ws_bytes, dev)
            B.check(B.lib().lidal_kmap_build(None, 0, None, n_out, None, k, int(self.symmetric),
                                             B.ptr(self.nbr_out), B.ptr(nbmaps), B.ptr(nbsizes),
                                             B.ptr(koff), 2, B.ptr(ws), ws_bytes, B.stream()),
                    'kmap_build(rules)')
            self._rules = (nbmaps, nbsizes, koff)
        return self._rules

    def streams(self):
        """(spairs i32 [cap, 2], sdesc i32, n_wg): the rule lists re-ordered into one stream per workgroup of the streamed
        weight gradient (lidal_wgrad_streams_build; stride-1 maps).  Built once per map, on the stream that is current at
        the first call -- prefetch_kernel_maps makes that call where the weight gradients will take this form, so the
        ~0.2 ms of table building sit with the other coordinate work."""
        if self._streams is None:
            k, n = self.nbr_out.shape
            dev = self.nbr_out.device
            L = B.lib()
            n_wg = int(L.lidal_wgrad_streams_workgroups())
            cap = int(L.lidal_wgrad_streams_rules(n, k, n_wg))
            spairs = B.empty((cap, 2), torch.int, dev)
            sdesc = torch.empty(int(L.lidal_wgrad_streams_desc_words(k, n_wg)), dtype=torch.int, device=dev)
            pairs, koff = self._nbmaps_cap, self.koff            # (before the workspace: they may have to be built first)
            key, key_k, key_range = self._stream_key if self._stream_key is not None else (None, 0, n)
            ws_bytes = int(L.lidal_wgrad_streams_workspace_bytes(n, k))
            ws = B.workspace(ws_bytes, dev)
            B.check(L.lidal_wgrad_streams_build(B.ptr(pairs), B.ptr(koff), k, n, B.ptr(key), key_k, key_range, n_wg,
                                                B.ptr(spairs), cap, B.ptr(sdesc), B.ptr(ws), ws_bytes, B.stream()),
                    'wgrad_streams_build')
            self._streams = (spairs, sdesc, n_wg)
        return self._streams

    def streams_serve(self, dtype, ca, cb, transposed=False):
        """Does the weight gradient of a [k, ca, cb] weight over this map take the streamed form?  bf16 operands, a
        stride-1 map of an odd kernel on at least backend.WGRAD_STREAMS_ROWS rows (the tables cost ~0.2 ms per map and
        step; below ~150 k rows the operands live in the L2s anyway), one channel tile."""
        n_in, n_out = self.sizes
        return bool(B.WGRAD_STREAMS_ROWS > 0 and dtype == torch.bfloat16 and not transposed and self.symmetric
                    and n_in == n_out and n_out >= B.WGRAD_STREAMS_ROWS and self.volume <= 32
                    and B.lib_handle().lidal_conv_wgrad_streams_serves(n_in, n_out, self.volume, ca, cb))

    @property
    def _nbmaps_cap(self):
        return self._build_rules()[0]

    @property
    def nbsizes(self):
        return self._build_rules()[1]

    @property
    def koff(self):
        return self._build_rules()[2]

    @property
    def total(self):
        if self._total is None:
            self._total = int(self.koff[-1].item())
        return self._total

    @property
    def nbmaps(self):
        """i32 [M, 2] = (in_idx, out_idx), grouped by offset, ascending out_idx (torchsparse)."""
        return self._nbmaps_cap[:self.total]

    @property
    def nbr_in(self):
        if self._nbr_in is None:
            n_in, n_out = self.sizes
            t = B.empty((self.volume, n_in), torch.int, self.nbr_out.device)
            B.check(B.lib().lidal_kmap_invert(B.ptr(self.nbr_out), n_out, self.volume, B.ptr(t),
                                              n_in, B.stream()), 'kmap_invert')
            self._nbr_in = t
        return self._nbr_in

    @property
    def order_out(self):
        if self._order_out is None:
            self._order_out = RowOrder(self.nbr_out)
        return self._order_out

    @property
    def order_in(self):
        if self._order_in is None:
            self._order_in = RowOrder(self.nbr_in)
        return self._order_in

    def __getitem__(self, i):                # torchsparse's [nbmaps, nbsizes, sizes] list view
        return (self.nbmaps, self.nbsizes, self.sizes)[i]


def build_kernel_map(coords, in_stride, kernel_size, stride, scope=None):
    """Cache-miss branch of upstream conv3d.  Returns (KernelMap, out_coords).  `scope` = the
    owning SparseTensor's cmaps dict (scopes the level's hash table, query.coords_table)."""
    B.require_gpu(coords)
    assert coords.dtype == torch.int
    coords = coords.contiguous()
    dev = coords.device
    offsets = get_kernel_offsets(kernel_size, stride=in_stride, device=dev)
    volume = offsets.shape[0]
    table = coords_table(coords, scope, in_stride)
    out_coords = coords
    if any(s > 1 for s in stride):
        # (prefetch_kernel_maps may have produced every level's coordinates already, from one sort)
        # -- only for kernel_size == stride: those are floor(c / s') s' of the input voxels, which is what the levels of
        # the pyramid hold; any other strided kernel has its own output set (torchsparse spdownsample) and must not pick up
        # a level that merely has the same stride (spdownsample raises for the shapes off the LiDAL path)
        out_stride = tuple(in_stride[k] * stride[k] for k in range(3))
        out_coords = scope.get(out_stride) if (scope is not None and tuple(kernel_size) == tuple(stride)) else None
        if out_coords is None:
            out_coords = spdownsample(coords, stride, kernel_size, in_stride)
    n_in, n_out = coords.shape[0], out_coords.shape[0]
    nbr_out = B.empty((volume, n_out), torch.int, dev)
    ws_bytes = B.lib().lidal_kmap_workspace_bytes(n_out, volume)
    ws = B.workspace(ws_bytes, dev)
    symmetric = (volume % 2 == 1) and all(s == 1 for s in stride)
    # training will ask for the rule lists (weight gradient): build them in the same call;
    # under no_grad only the neighbour table is built (KernelMap derives the rest if ever asked)
    rules = None
    if torch.is_grad_enabled():
        rules = (B.empty((volume * n_out, 2), torch.int, dev),
                 torch.empty(volume, dtype=torch.int, device=dev),
                 torch.empty(volume + 1, dtype=torch.int64, device=dev))
    B.check(B.lib().lidal_kmap_build(B.ptr(table.buf), table.nbytes, B.ptr(out_coords), n_out,
                                     B.ptr(offsets), volume, int(symmetric), B.ptr(nbr_out),
                                     B.ptr(rules[0]) if rules else None,
                                     B.ptr(rules[1]) if rules else None,
                                     B.ptr(rules[2]) if rules else None,
                                     0 if rules else 1, B.ptr(ws), ws_bytes, B.stream()),
            'kmap_build')
    kmap = KernelMap(nbr_out, (n_in, n_out), volume, symmetric)
    kmap._rules = rules
    return kmap, out_coords


def build_kernel_maps(jobs, scope):
    """Several kernel maps in one chain of launches (lidal_kmap_build_batch).  jobs: list of
    (in_coords, in_stride, kernel_size, stride, out_coords) with out_coords given (the level's own coordinates,
    or those of the coarser level: F.downsample_pyramid).  -> list of KernelMap, each what build_kernel_map
    returns for the same arguments."""
    import ctypes
    if not jobs:
        return []
    dev = jobs[0][0].device
    want_rules = torch.is_grad_enabled()
    rows = []
    for coords, in_stride, kernel_size, stride, out_coords in jobs:
        assert coords.dtype == torch.int and out_coords.dtype == torch.int
        coords, out_coords = coords.contiguous(), out_coords.contiguous()
        offsets = get_kernel_offsets(kernel_size, stride=in_stride, device=dev)
        volume = offsets.shape[0]
        table = coords_table(coords, scope, in_stride)
        n_in, n_out = coords.shape[0], out_coords.shape[0]
        nbr_out = B.empty((volume, n_out), torch.int, dev)
        symmetric = (volume % 2 == 1) and all(s_ == 1 for s_ in stride)
        rules = None
        if want_rules:
            rules = (B.empty((volume * n_out, 2), torch.int, dev),
                     torch.empty(volume, dtype=torch.int, device=dev),
                     torch.empty(volume + 1, dtype=torch.int64, device=dev))
        rows.append((table, out_coords, offsets, volume, symmetric, nbr_out, rules, n_in, n_out))
    out = []
    for c0 in range(0, len(rows), 12):
        chunk = rows[c0:c0 + 12]
        n = len(chunk)
        vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
        n_out_a, k_a = i64(*[r[8] for r in chunk]), i32(*[r[3] for r in chunk])
        L = B.lib()
        ws_bytes = L.lidal_kmap_build_batch_workspace_bytes(n_out_a, k_a, n)
        ws = B.workspace(ws_bytes, dev)
        B.check(L.lidal_kmap_build_batch(
            vp(*[r[0].buf.data_ptr() for r in chunk]), i64(*[r[0].nbytes for r in chunk]),
            vp(*[r[1].data_ptr() for r in chunk]), n_out_a, vp(*[r[2].data_ptr() for r in chunk]), k_a,
            i32(*[int(r[4]) for r in chunk]), vp(*[r[5].data_ptr() for r in chunk]),
            vp(*[(r[6][0].data_ptr() if r[6] else None) for r in chunk]),
            vp(*[(r[6][1].data_ptr() if r[6] else None) for r in chunk]),
            vp(*[(r[6][2].data_ptr() if r[6] else None) for r in chunk]), n, B.ptr(ws), ws_bytes, B.stream()),
            'kmap_build')
        for table, out_coords, offsets, volume, symmetric, nbr_out, rules, n_in, n_out in chunk:
            kmap = KernelMap(nbr_out, (n_in, n_out), volume, symmetric)
            kmap._rules = rules
            out.append(kmap)
    return out


def prefetch_kernel_maps(x, plan, transposed=True):
    """Build, ahead of the feature kernels, every coordinate set and kernel map a network will ask
    for: `plan` is the sequence of (kernel_size, stride) of its non-transposed convs along the
    encoder, starting at x's stride.  Each entry lands in x.cmaps / x.kmaps under exactly the key
    conv3d looks up, so nothing changes downstream -- except that the host waits for the
    data-dependent output sizes (one per strided conv) happen here, while the GPU queue is still
    short, instead of draining the queue in the middle of the network.  `transposed` also
    prepares the transposed tables the decoder uses."""
    coords, cur = x.coords, tuple(x.stride)
    x.cmaps.setdefault(cur, coords)
    # the coordinates of every coarser level from ONE sort (F.downsample_pyramid) when the encoder is the usual
    # chain of stride-2 / kernel-2 downsamplings: one host round trip instead of one per level
    downs = [(make_ntuple(k, ndim=3), make_ntuple(s_, ndim=3)) for k, s_ in plan
             if any(v > 1 for v in make_ntuple(s_, ndim=3))]
    if (1 <= len(downs) <= 4 and all(k == (2, 2, 2) and s_ == (2, 2, 2) for k, s_ in downs)
            and coords.is_cuda and coords.shape[0] > 0 and max(cur) << len(downs) < 65536):
        strides = [tuple(c << (l + 1) for c in cur) for l in range(len(downs))]
        if not any(st in x.cmaps for st in strides):
            from .downsample import downsample_pyramid
            try:
                for st, c in zip(strides, downsample_pyramid(coords, len(downs), cur)):
                    x.cmaps[st] = c
            except ValueError:          # a batch index >= 8192 / coordinates the packed key cannot hold: the level-by-level
                pass                    # path below (F.spdownsample: batch < 32768) takes over
    # ... and with every level's coordinates known, all kernel maps in one chain of launches
    todo, keys_todo = [], []
    c_, cur_ = coords, cur
    for kernel_size, stride in plan:
        kernel_size = make_ntuple(kernel_size, ndim=3)
        stride = make_ntuple(stride, ndim=3)
        if kernel_size == (1, 1, 1) and stride == (1, 1, 1):
            continue
        key = (cur_, kernel_size, stride, (1, 1, 1))
        out_stride = tuple(cur_[k] * stride[k] for k in range(3))
        strided = any(s_ > 1 for s_ in stride)
        out_c = x.cmaps.get(out_stride) if strided else c_
        if out_c is None or c_.shape[0] == 0 or out_c.shape[0] == 0 or not c_.is_cuda:
            todo = []
            break                                   # a level without coordinates yet: the one-by-one path below
        if key not in x.kmaps and key not in keys_todo:
            todo.append((c_, cur_, kernel_size, stride, out_c))
            keys_todo.append(key)
        if strided:
            c_, cur_ = out_c, out_stride
    if len(todo) > 1:
        for key, kmap in zip(keys_todo, build_kernel_maps(todo, x.cmaps)):
            x.kmaps[key] = kmap
    want = []                   # (kmap, 'out' | 'in') whose row order is still to be built
    for kernel_size, stride in plan:
        kernel_size = make_ntuple(kernel_size, ndim=3)
        stride = make_ntuple(stride, ndim=3)
        if kernel_size == (1, 1, 1) and stride == (1, 1, 1):
            continue
        key = (cur, kernel_size, stride, (1, 1, 1))
        out_stride = tuple(cur[k] * stride[k] for k in range(3))
        kmap = x.kmaps.get(key)
        if kmap is None:
            kmap, out_coords = build_kernel_map(coords, cur, kernel_size, stride, x.cmaps)
            x.kmaps[key] = kmap
            if any(s > 1 for s in stride):
                x.cmaps.setdefault(out_stride, out_coords)
        if kmap._order_out is None:
            want.append((kmap, 'out'))
        if any(s > 1 for s in stride):
            if transposed and kmap._order_in is None:
                want.append((kmap, 'in'))
            coords, cur = x.cmaps[out_stride], out_stride
    if want:                    # all row orders of the network: one sort per kernel volume
        orders = RowOrder.build_many([km.nbr_out if side == 'out' else km.nbr_in for km, side in want])
        for (km, side), o in zip(want, orders):
            if side == 'out':
                km._order_out = o
            else:
                km._order_in = o
    if torch.is_grad_enabled() and B.WGRAD_STREAMS_ROWS > 0:
        # the stream tables of the weight gradients on the large levels (KernelMap.streams).  The rows of the level the
        # network enters on are numbered as the caller numbered them (SPVCNN: by coordinate hash): their spatial key is
        # the parent row on the next level -- the strided map's inverse neighbour table; every coarser level is numbered
        # in coordinate order (F.spdownsample) and keyed by its row index.
        first = tuple(x.stride)
        for key, km in x.kmaps.items():
            if (key[1] == (3, 3, 3) and key[2] == (1, 1, 1) and km.symmetric and km._streams is None
                    and km.sizes[0] >= B.WGRAD_STREAMS_ROWS and km.nbr_out.is_cuda):
                if key[0] == first and km._stream_key is None:
                    down = x.kmaps.get((first, (2, 2, 2), (2, 2, 2), (1, 1, 1)))
                    if down is not None and down.sizes[0] == km.sizes[0]:
                        km._stream_key = (down.nbr_in, down.volume, down.sizes[1])
                km.streams()
    return x


def _weight_image(weight, dtype, n_out, role, code=None):
    """LDS image (csrc/conv_img.hip) of a [K, ci, co] weight in `dtype` for a convolution that
    produces n_out rows.  role 0: forward operand (reduction over ci, columns co); role 1: the
    data-gradient operand of the same parameter (reduction over co, columns ci).
    Outside autograd (inference: the weights do not change between calls) the image is cached on the
    weight tensor, keyed by its version counter, storage, role, dtype and the tiling n_out selects."""
    k, ci, co = weight.shape
    n_red, n_col = (ci, co) if role == 0 else (co, ci)
    code = B.dtype_code(dtype) if code is None else code        # (B.F32_SPLIT: the split image of an f32 product)
    L = B.lib()
    key = cache = None
    if not torch.is_grad_enabled():
        key = (B.weights_key(weight), role, code, _tiling(n_red, n_col, code, n_out))
        cache = getattr(weight, '_lidal_images', None)
        if cache is not None and key in cache:
            return cache[key]
    w = weight.detach().contiguous()
    nbytes = L.lidal_conv_weight_image_bytes(k, n_red, n_col, code, n_out)
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    B.check(L.lidal_conv_weight_image(B.ptr(w), B.dtype_code(w.dtype), role, B.ptr(img), code,
                                      k, n_red, n_col, n_out, B.stream()), 'conv_weight_image')
    if key is not None:
        if cache is None or next(iter(cache))[0] != B.weights_key(weight):      # drop images of older versions
            cache = {}
            weight._lidal_images = cache
        cache[key] = img
    return img


_TILING = {}


def _tiling(n_red, n_col, code, n_out):
    """lidal_conv_weight_image_tiling, memoised: a pure function, asked ~100 times per step."""
    key = (n_red, n_col, code, n_out)
    t = _TILING.get(key)
    if t is None:
        if len(_TILING) > 4096:
            _TILING.clear()
        t = _TILING[key] = B.lib().lidal_conv_weight_image_tiling(n_red, n_col, code, n_out)
    return t


class _ImageBank:
    """Training: the image pairs of every convolution weight, rebuilt by ONE launch per step.
    A weight registers itself at its first use (its images then live in a persistent buffer, keyed by
    the tilings the two row counts select); whenever a requested image is older than its weight
    (backend.weights_key: a backward pass has ended since, or the version counter moved), ALL
    registered weights of that device / dtype whose
    images are stale are rebuilt together (lidal_conv_weight_image_batch) -- 42 launches of ~7 us
    become one.  LIDAL_IMAGE_BATCH=0 rebuilds per call as before.
    A write to a parameter that neither moves its version counter nor follows a backward pass (e.g.
    through `param.data` in a pure inference process) is invisible to every cache of this package:
    call `backend._bump_epoch()` after it."""

    def __init__(self):
        self.entries = {}           # (id(weight), key) -> entry dict; key = (storage, dtypes, the two tilings, role)
        self.serial = 0             # entries are told apart by a serial number (id()s get re-used)
        self.tables = {}            # (device, code, w_code) -> (signature, device table, n_jobs, total)
    KEEP = 8                        # an entry nobody asked for during the last KEEP weight epochs (backward passes) is dropped

    @property
    def tick(self):
        return B.WEIGHT_EPOCH[0]

    def get(self, weight, dtype, n_out_fwd, n_out_bwd, shape=None, role=0, code=None):
        """`shape` = (k, ci, co) of the operand when `weight` is not [k, ci, co] itself: a [ci, co]
        1x1x1 kernel (role 0) or nn.Linear's [co, ci] weight (role 1).
        A weight may have several entries at a time, one per pair of tilings its row counts have selected
        recently (sizes that change from batch to batch straddle the tiling thresholds of a few layers): all of
        them ride in the same launch, none is rebuilt for being asked under another tiling."""
        import weakref
        k, ci, co = shape if shape is not None else weight.shape
        code = B.dtype_code(dtype) if code is None else code
        L = B.lib()
        key = (weight.data_ptr(), code, weight.dtype, _tiling(ci, co, code, n_out_fwd), _tiling(co, ci, code, n_out_bwd), role)
        ekey = (id(weight), key)
        e = self.entries.get(ekey)
        if e is None or e['ref']() is not weight:
            nf = L.lidal_conv_weight_image_bytes(k, ci, co, code, n_out_fwd)
            # (the split form: a data-gradient image where its reduction -- co -- is whole 32-channel slices too: training)
            nb = (0 if (code == B.F32_SPLIT and (co % 32 != 0 or role))
                  else L.lidal_conv_weight_image_bytes(k, co, ci, code, n_out_bwd))
            buf = torch.empty(nf + nb, dtype=torch.uint8, device=weight.device)
            self.serial += 1
            wid = id(weight)

            def gone(_r, wid=wid):
                for kk in [kk for kk in self.entries if kk[0] == wid]:
                    self.entries.pop(kk, None)
            e = {'ref': weakref.ref(weight, gone), 'key': key,
                 'serial': self.serial,
                 'buf': buf, 'img_f': buf[:nf], 'img_b': buf[nf:], 'version': -1, 'code': code,
                 'n_out': (n_out_fwd, n_out_bwd), 'shape': (k, ci, co), 'role': role, 'used': self.tick,
                 'group': (str(weight.device), code, B.dtype_code(weight.dtype))}
            self.entries[ekey] = e
        e['used'] = self.tick
        if e['version'] != B.weights_key(weight):
            self._rebuild(e['group'])
        return e['img_f'], e['img_b']

    def entry(self, weight, tkey):
        """The entry of `weight` under the tilings / dtype `tkey` = (tiling_fwd, tiling_bwd, code), or None."""
        for kk, e in self.entries.items():
            if kk[0] == id(weight) and (kk[1][3], kk[1][4], kk[1][1]) == tkey and e['ref']() is weight:
                return e
        return None

    def _rebuild(self, group):
        """All stale images of `group` in one launch."""
        L = B.lib()
        stale = []
        for kk in [kk for kk, e in self.entries.items() if e['used'] < self.tick - self.KEEP]:
            del self.entries[kk]
        for e in self.entries.values():
            w = e['ref']()
            if w is not None and e['group'] == group and e['version'] != B.weights_key(w):
                stale.append((e, w))
        if not stale:
            return
        sig = tuple(e['serial'] for e, _ in stale)
        cached = self.tables.get(group)
        if cached is None or cached[0] != sig:
            import ctypes
            jb = L.lidal_conv_weight_image_job_bytes()
            host = ctypes.create_string_buffer(jb * len(stale))
            first = 0
            for i, (e, w) in enumerate(stale):
                k, ci, co = e['shape']
                n = L.lidal_conv_weight_image_job(ctypes.c_void_p(ctypes.addressof(host) + i * jb), B.ptr(w),
                                                  e['role'], B.ptr(e['img_f']), e['n_out'][0],
                                                  B.ptr(e['img_b']) if e['img_b'].numel() else None,
                                                  e['n_out'][1], e['code'], k, ci, co, first)
                if n < 0:
                    B.check(1, 'conv_weight_image_job')
                first += n
            table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(stale[0][1].device)
            cached = (sig, table, len(stale), first)
            self.tables[group] = cached
        _, table, n_jobs, total = cached
        B.check(L.lidal_conv_weight_image_batch(B.ptr(table), n_jobs, total, group[2], group[1], B.stream()),
                'conv_weight_image')
        for e, w in stale:
            e['version'] = B.weights_key(w)


_IMAGE_BANK = _ImageBank()
_IMAGE_BATCH = os.environ.get('LIDAL_IMAGE_BATCH', '1') != '0'


def _weight_image_pair(weight, dtype, n_out_fwd, n_out_bwd, code=None):
    """(forward image, data-gradient image) of a [K, ci, co] weight: from the step's one batched
    launch (_ImageBank) for contiguous parameters, else from ONE launch of their own."""
    if _IMAGE_BATCH and isinstance(weight, torch.nn.Parameter) and weight.is_contiguous():
        return _IMAGE_BANK.get(weight, dtype, n_out_fwd, n_out_bwd, code=code)
    k, ci, co = weight.shape
    w = weight.detach().contiguous()
    if code == B.F32_SPLIT:             # (no pair entry point for the split form: one launch per image)
        with torch.enable_grad():       # (bypass the per-tensor inference cache)
            return (_weight_image(w, dtype, n_out_fwd, 0, code), _weight_image(w, dtype, n_out_bwd, 1, code))
    code = B.dtype_code(dtype)
    L = B.lib()
    nf = L.lidal_conv_weight_image_bytes(k, ci, co, code, n_out_fwd)
    nb = L.lidal_conv_weight_image_bytes(k, co, ci, code, n_out_bwd)
    buf = torch.empty(nf + nb, dtype=torch.uint8, device=w.device)
    img_f, img_b = buf[:nf], buf[nf:]
    B.check(L.lidal_conv_weight_image_pair(B.ptr(w), B.dtype_code(w.dtype), B.ptr(img_f), n_out_fwd,
                                           B.ptr(img_b), n_out_bwd, code, k, ci, co, B.stream()),
            'conv_weight_image')
    return img_f, img_b


def _apply(feats, img, k, co, order, kflip, epilogue=None, want_stats=False, bnb=None, code=None):
    """out[j] = sum_k feats[nbr[kk][j]] @ W_k with the weights as the LDS image `img` (built for
    (ci = feats.shape[1], co, k, feats.dtype, n_out = order.n_rows)); `order` = RowOrder(nbr).
    epilogue = (scale f32 [co], shift f32 [co], relu[, residual [n_out, co]]): in-kernel
    out = act(out * scale + shift) + residual; relu 1 = ReLU before the sum, 2 = after it."""
    ci = feats.shape[1]
    n_out = order.n_rows
    out = torch.empty((n_out, co), dtype=feats.dtype, device=feats.device)
    scale, shift, relu = epilogue[:3] if epilogue is not None else (None, None, 0)
    residual = epilogue[3] if epilogue is not None and len(epilogue) > 3 else None
    if residual is not None:
        residual = residual.contiguous().to(feats.dtype)
        assert residual.shape == (n_out, co)
    stats = None
    # bf16 only: the tile triples are f32, and in the f32 parity mode the BatchNorm statistics keep
    # their own f64-accumulated pass (an ill-conditioned gamma gradient of the golden model notices)
    want_stats = want_stats and feats.dtype == torch.bfloat16
    if want_stats:              # (count, mean, M2) per 128-row tile and column, for the BatchNorm that follows
        stats = torch.empty((-(-n_out // B.stats_tile_rows()), co, 3), dtype=torch.float32, device=feats.device)
    # the library may split the offsets of a coarse level's tiles over several workgroups (csrc/conv_img.hip, Split):
    # the f32 partial tiles go through a workspace
    ws_bytes = apply_workspace_bytes(n_out, co)
    ws = B.workspace(ws_bytes, feats.device) if ws_bytes else None
    if bnb is not None:         # a data gradient that also leaves the backward sums of the BatchNorm in front of it
        bx, mean, invstd, gamma, beta, relu_bn = bnb
        assert epilogue is None and not want_stats and bx.shape == (n_out, co) and bx.dtype == feats.dtype
        sums = torch.empty((-(-n_out // B.stats_tile_rows()), co, 2), dtype=torch.float32, device=feats.device)
        B.check(B.lib().lidal_conv_dgrad_bn_sums_ws(B.ptr(feats), B.ptr(img), B.ptr(order.table), B.ptr(order.perm),
                                                    B.ptr(order.tile_masks), B.ptr(out), feats.shape[0], n_out, ci,
                                                    co, k, int(kflip), B.dtype_code(feats.dtype), B.ptr(bx),
                                                    B.ptr(mean), B.ptr(invstd), B.ptr(gamma), B.ptr(beta),
                                                    int(bool(relu_bn)), B.ptr(sums), B.ptr(ws), ws_bytes, B.stream()),
                'conv_apply')
        out._lidal_bnb_sums = sums
        return out
    B.check(B.lib().lidal_conv_apply_image_ws(B.ptr(feats), B.ptr(img), B.ptr(order.table), B.ptr(order.perm),
                                              B.ptr(order.tile_masks), B.ptr(out), feats.shape[0], n_out, ci,
                                              co, k, int(kflip), B.dtype_code(feats.dtype) if code is None else code,
                                              B.ptr(scale), B.ptr(shift), int(relu), B.ptr(residual), B.ptr(stats),
                                              B.ptr(ws), ws_bytes, B.stream()),
            'conv_apply')
    if want_stats:
        out._lidal_bn_stats = stats     # picked up by spnn.BatchNorm (nn/functional/norm.py)
    return out


_APPLY_WS = {}
_SPLIT = os.environ.get('LIDAL_CONV_SPLIT', '1') != '0'


def apply_workspace_bytes(n_out, co):
    """lidal_conv_apply_workspace_bytes, memoised (0 with LIDAL_CONV_SPLIT=0: no launch is split then)."""
    if not _SPLIT:
        return 0
    key = (n_out, co)
    v = _APPLY_WS.get(key)
    if v is None:
        if len(_APPLY_WS) > 4096:
            _APPLY_WS.clear()
        v = _APPLY_WS[key] = int(B.lib_handle().lidal_conv_apply_workspace_bytes(n_out, co))
    return v


def wgrad_scratch(n_a, n_b, k, ca, cb, dtype, device, code=None):
    """The f32 [slabs, ca, cb] scratch lidal_conv_wgrad reduces through; the slab count (workgroups
    of the launch + k for the bf16 DMA kernel, split-K slabs x k otherwise; with B.F32_SPLIT also the room
    for the operands' bf16 pieces) is the library's plan."""
    slabs = int(B.lib().lidal_conv_wgrad_slabs(n_a, n_b, k, ca, cb, B.dtype_code(dtype) if code is None else code))
    return torch.empty((slabs, ca, cb), dtype=torch.float32, device=device)


def wgrad_plan(n_a, n_b, k, ca, cb, dtype):
    """(dtype code, slabs) of a weight gradient: the split form for f32 operands where the library serves the shape
    (backend.wgrad_code; lidal_conv_wgrad_slabs says -1 otherwise), else the operands' own dtype.  One rule for the
    per-operator path and the planned step."""
    code = B.wgrad_code(dtype, ca, cb)
    L = B.lib_handle()
    if code == B.F32_SPLIT:
        slabs = int(L.lidal_conv_wgrad_slabs(n_a, n_b, k, ca, cb, code))
        if slabs > 0:
            return code, slabs
        code = B.dtype_code(dtype)
    return code, int(L.lidal_conv_wgrad_slabs(n_a, n_b, k, ca, cb, code))


def _pad_channels(ci, dtype):
    """Input channels to append so that a row is whole 16-byte vectors (the 4-channel stem in bf16:
    8-byte rows would go through the element-wise guarded loads, 3x slower)."""
    vec = 8 if dtype == torch.bfloat16 else 4
    return (-ci) % vec


def _bwd_order(kmap, transposed):
    """(RowOrder, kflip) of the data gradient of a convolution over `kmap`."""
    if transposed:
        return kmap.order_out, 0
    return (kmap.order_out, 1) if kmap.symmetric else (kmap.order_in, 0)


def _forward(feats, weight, kmap, transposed, epilogue=None, with_bwd_image=False, want_stats=False, inference=False):
    """-> (x in the compute dtype, out, image of the data gradient or None).  `inference` (no autograd node will be
    made): f32 products run in the split form (backend.conv_code)."""
    B.require_gpu(feats, weight)
    cdtype = B.compute_dtype(feats)
    x = feats.contiguous().to(cdtype)
    pad = _pad_channels(x.shape[1], cdtype)
    if pad:                     # zero channels times zero weight rows: same sums
        x = torch.nn.functional.pad(x, (0, pad))
        weight = torch.nn.functional.pad(weight.detach(), (0, 0, 0, pad))
    order = kmap.order_in if transposed else kmap.order_out
    k, _, co = weight.shape
    img_b = None
    # (training: the split form where both channel counts allow it -- the data gradient reduces over co; backend.conv_code)
    code = B.conv_code(cdtype, x.shape[1], inference and not want_stats, None if inference else co)
    if with_bwd_image:          # both operands of this parameter from one launch
        img, img_b = _weight_image_pair(weight, cdtype, order.n_rows, _bwd_order(kmap, transposed)[0].n_rows, code)
        img_b._lidal_code = code        # (conv_backward multiplies with the image as it was built)
        return x, _apply(x, img, k, co, order, 0, epilogue, want_stats, None, code), img_b
    img = _weight_image(weight, cdtype, order.n_rows, 0, code)
    return x, _apply(x, img, k, co, order, 0, epilogue, want_stats, None, code), img_b


def _conv(feats, weight, kmap, transposed, epilogue=None, want_stats=False, fork=False):
    """`fork`: return (out, alias of `feats`) -- the alias is for a second consumer (the shortcut of a
    residual block); the gradient arriving through it is added to the data gradient inside the
    data-gradient kernel's epilogue instead of by a separate pass (autograd's accumulation)."""
    if B.wants_grad(feats, weight):
        assert epilogue is None, 'the fused BatchNorm epilogue is inference-only'
        # (the fused sum needs the skip gradient in the data gradient's own width: no channel padding)
        fused = fork and feats.requires_grad and _pad_channels(feats.shape[1], B.compute_dtype(feats)) == 0
        res = ConvolutionFunction.apply(feats, weight, kmap, transposed, want_stats, fused)
        out, skip = res if fused else (res, feats)
        if ConvolutionFunction.last_stats is not None:      # from the Function's own output to autograd's
            out._lidal_bn_stats = ConvolutionFunction.last_stats
            ConvolutionFunction.last_stats = None
        return (out, skip) if fork else out
    out = _forward(feats, weight, kmap, transposed, epilogue, False, want_stats, True)[1]   # no autograd node
    return (out, feats) if fork else out


def conv_backward(x, weight, kmap, transposed, img_bwd, grad_output, grad_skip=None, need_gx=True, need_gw=True,
                  bnb=None):
    """Backward of a sparse convolution on raw tensors: x = the (compute-dtype, channel-padded) input
    _forward returned, img_bwd = its data-gradient image.  -> (grad_in or None, grad_w or None).
    `grad_skip`: a gradient reaching the input through a second consumer, added in the data-gradient
    kernel's epilogue.  `bnb` = (bn_x, mean, invstd, gamma, beta, relu): the input was act(bn(bn_x)) and has
    no other consumer -- the data-gradient launch then also leaves that BatchNorm's backward sums per tile
    on grad_in (`_lidal_bnb_sums`, for norm.train_backward): bf16, no channel padding, no grad_skip.
    Shared by ConvolutionFunction and the fused block Functions of lidal_amd.network."""
    kmap.check_alive()
    g = grad_output.contiguous().to(x.dtype)
    n_in, n_out = kmap.sizes
    grad_in = grad_w = None
    k, ci_w, co = weight.shape
    ci = x.shape[1]                          # >= ci_w when the input was channel-padded

    def wgrad():
        gw = torch.empty((k, ci, co), dtype=torch.float32, device=x.device)
        if kmap.streams_serve(x.dtype, ci, co, transposed):          # large stride-1 levels: one rule stream per workgroup
            spairs, sdesc, n_wg = kmap.streams()
            partial = torch.empty((2 * n_wg, ci, co), dtype=torch.float32, device=x.device)
            B.check(B.lib().lidal_conv_wgrad_streams(B.ptr(x), B.ptr(g), x.shape[0], g.shape[0], B.ptr(spairs), B.ptr(sdesc),
                                                     n_wg, 0, B.ptr(gw), B.ptr(partial), partial.shape[0], k, ci, co,
                                                     B.BF16, B.stream()), 'conv_wgrad_streams')
            gw = gw[:, :ci_w].contiguous() if ci != ci_w else gw
            return gw if weight.dtype == torch.float32 else gw.to(weight.dtype)
        code, slabs = wgrad_plan(x.shape[0], g.shape[0], k, ci, co, x.dtype)
        partial = torch.empty((slabs, ci, co), dtype=torch.float32, device=x.device)
        B.check(B.lib().lidal_conv_wgrad(B.ptr(x), B.ptr(g), x.shape[0], g.shape[0],
                                         B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff),
                                         1 if transposed else 0, B.ptr(gw), B.ptr(partial),
                                         partial.shape[0], k, ci, co, code,
                                         B.stream()), 'conv_wgrad')
        gw = gw[:, :ci_w].contiguous() if ci != ci_w else gw
        return gw if weight.dtype == torch.float32 else gw.to(weight.dtype)

    side = None
    if need_gw:
        if B.overlap_wgrad(x.dtype, max(n_in, n_out)) and need_gx:
            _ = kmap.koff                   # the rule lists are built on the main stream
            side = B.beside(x.device, (x, g, kmap._nbmaps_cap, kmap.koff)
                            + (tuple(kmap.streams()[:2]) if kmap.streams_serve(x.dtype, ci, co, transposed) else ()), weight)
            with side as done:              # beside the data gradient below
                grad_w = wgrad()
                done(grad_w)
        else:
            grad_w = wgrad()
    if need_gx:
        # gin[i] = sum_k gout[.] @ W[k]^T : "output channels" are ci (of the padded input), reduction
        # over co; the operand image was built together with the forward one
        order, kflip = _bwd_order(kmap, transposed)
        # the gradient of the forked alias joins in the epilogue (out = acc + residual)
        ep = (None, None, 0, grad_skip) if grad_skip is not None else None
        if bnb is not None and (ep is not None or ci != ci_w or g.dtype != torch.bfloat16 or ci % 8 != 0
                                or tuple(bnb[0].shape) != (n_out if transposed else n_in, ci)
                                or bnb[0].dtype != g.dtype or not bnb[0].is_contiguous()):
            bnb = None
        grad_in = _apply(g, img_bwd, k, x.shape[1], order, kflip, ep, False, bnb, getattr(img_bwd, '_lidal_code', None))
        if ci != ci_w:
            grad_in = grad_in[:, :ci_w]                 # drop the padding channels, if any
    elif grad_skip is not None:
        grad_in = grad_skip
    if side is not None:
        side.finish()
    return grad_in, grad_w


class ConvolutionFunction(Function):
    last_stats = None           # tile statistics of the latest forward (a non-differentiable side output)

    @staticmethod
    def forward(ctx, feats, weight, kmap, transposed, want_stats=False, fork=False):
        x, out, ctx.img_bwd = _forward(feats, weight, kmap, transposed, None, ctx.needs_input_grad[0],
                                       want_stats)
        ConvolutionFunction.last_stats = getattr(out, '_lidal_bn_stats', None) if want_stats else None
        ctx.kmap = kmap
        ctx.transposed = transposed
        ctx.save_for_backward(x, weight)
        return (out, feats) if fork else out          # an input returned as is becomes an alias of it

    @staticmethod
    def backward(ctx, grad_output, grad_skip=None):
        B.note_backward()
        x, weight = ctx.saved_tensors
        grad_in, grad_w = conv_backward(x, weight, ctx.kmap, ctx.transposed, ctx.img_bwd, grad_output, grad_skip,
                                        ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return grad_in, grad_w, None, None, None, None


def conv3d(input, weight, kernel_size, bias=None, stride=1, dilation=1, transposed=False,
           epilogue=None, want_stats=False, fork=False):
    """torchsparse F.conv3d.  `epilogue` (inference only, not for 1x1x1 kernels) = (scale, shift,
    relu) applies the per-channel affine map of a following eval-mode BatchNorm (+ ReLU) inside
    the convolution kernel.  `want_stats` (training: a train-mode BatchNorm follows) makes the kernel
    leave that BatchNorm's batch statistics, reduced per 128-row tile, on the output features.
    `fork` (k > 1, not transposed): returns (output, alias of input) -- see _conv."""
    feats, coords = input.feats, input.coords
    kernel_size = make_ntuple(kernel_size, ndim=3)
    stride = make_ntuple(stride, ndim=3)
    dilation = make_ntuple(dilation, ndim=3)

    if kernel_size == (1, 1, 1) and stride == (1, 1, 1) and dilation == (1, 1, 1):
        B.require_gpu(feats)
        from .dense import rows_matmul
        feats = rows_matmul(feats, weight, bias, epilogue, want_stats)
        output = SparseTensor(feats, coords, input.stride)
    elif not transposed:
        if dilation != (1, 1, 1):
            raise NotImplementedError('dilated sparse conv is not on the LiDAL path')
        key = (input.stride, kernel_size, stride, dilation)
        kmap = input.kmaps.get(key)
        trace = getattr(input.kmaps, 'trace', None)
        if trace is not None and key[:3] not in trace['maps']:
            trace['maps'].append(key[:3])        # (nn.Conv3d: what THIS forward pass asks for, for the next one's prefetch)
        out_stride = tuple(input.stride[k] * stride[k] for k in range(3))
        if kmap is None:
            kmap, out_coords = build_kernel_map(coords, input.stride, kernel_size, stride, input.cmaps)
            input.kmaps[key] = kmap
            if any(s > 1 for s in stride):
                input.cmaps.setdefault(out_stride, out_coords)
        out_coords = coords if all(s == 1 for s in stride) else input.cmaps[out_stride]
        if fork:
            feats, skip_feats = _conv(feats, weight, kmap, False, epilogue, want_stats and bias is None, True)
        else:
            feats = _conv(feats, weight, kmap, False, epilogue, want_stats and bias is None)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(feats, out_coords, out_stride)
    else:
        tensor_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        kmap = input.kmaps[(tensor_stride, kernel_size, stride, dilation)]
        trace = getattr(input.kmaps, 'trace', None)
        if trace is not None:
            trace['transposed'] = True
        feats = _conv(feats, weight, kmap, True, epilogue, want_stats and bias is None)
        if bias is not None:
            feats = feats + bias
        output = SparseTensor(feats, input.cmaps[tensor_stride], tensor_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    if fork:
        assert kernel_size != (1, 1, 1) and not transposed, 'fork: k > 1, not transposed'
        skip = SparseTensor(skip_feats, coords, input.stride)
        skip.cmaps, skip.kmaps = input.cmaps, input.kmaps
        return output, skip
    return output
