"""The coordinate-only part of a forward pass, built ahead of the features -- optionally on a second stream.

Everything SPVCNN / MinkUNet derive from the voxel coordinates alone -- the re-voxelisation index
(network/utils.py:14-31), the coordinates of the four coarser levels, the nine kernel maps with their row
orders and rule lists (torchsparse conv3d's cache-miss branch), the point <-> voxel indices, trilinear weights
and contributor lists at strides 1 / 16 / 4 (network/utils.py:39-53,67-92) -- depends on neither the features
nor the weights.  The reference builds all of it inside `model(x)`, every iteration, and so does this package by
default.  Inside the forward pass it costs more than its kernels: the sizes of the data-dependent outputs
(number of voxels, rows per level, rules per map) come back to the host through three stream synchronisations,
each of which first drains everything queued before it -- the previous step's backward pass, or the previous
frame's forward -- and then leaves the GPU idle while the host queues the next small launches.

`Geometry.build(model, coords)` runs that part by itself; `GeometryPrefetcher.submit(coords)` runs it on a second
HIP stream, so its synchronisations wait only for its own short queue while the main stream keeps working on
the previous batch (the reference overlaps the same way one stage earlier: its DataLoader workers voxelise the
next scans on the CPU while the GPU trains, dataset/sk_dataloader.py:21,53).  Hand the result to the forward pass
with `x.geometry = g` (train_step / infer_frame take it as `geometry=`): the model finds every table in its
caches and launches feature kernels only.  Same tables, bit for bit, as the in-line path builds
(tests/test_geometry_gpu.py); every batch still gets its own build.
"""
import torch

from .. import PointTensor, SparseTensor
from ..nn import functional as F
from ..nn.functional.conv import prefetch_kernel_maps
from ..nn.functional.devoxelize import prepare_devoxelize
from ..nn.functional.voxelize import _index32
from ..nn.functional.voxelize import prepare_voxelize
from .glue import corner_tables, initial_tables, point_tables

__all__ = ['Geometry', 'GeometryPrefetcher']


class Geometry:
    """What one forward pass of `kind` ('SPVCNN' / 'MinkUNet') derives from `coords` (i32 [N,4], batch last)."""

    def __init__(self, coords, kind, grad):
        self.coords = coords
        self.kind = kind
        self.grad = grad                # rule lists / backward contributor lists included
        self.x0 = None                  # SparseTensor without features: level-0 coordinates + cmaps / kmaps
        self.z = None                   # SPVCNN: PointTensor without features carrying the point caches
        self.ready = None               # event on the stream the tables were built on (None: the caller's own stream)
        self._stream = None
        self._consumer = None           # the stream whose kernels read the tables (fenced by the prefetcher)
        self.payload = None             # GeometryPrefetcher.submit_batch: the input batch built on the same stream
        self._arena = None              # backend.BlockArena the tables were carved from
        self._age = 0                   # submissions to the same prefetcher since this one

    @staticmethod
    def build(model, coords, grad=None, arena=None):
        """Build on the current stream.  grad: also what only a backward pass reads (default: model.training and
        gradients enabled).  The tables are carved out of the blocks of one backend.BlockArena (`arena`, or a new
        one): same-sized blocks the allocator re-uses exactly, whatever this batch's voxel counts are."""
        from .unet import SPVCNN, MinkUNet
        from .. import backend as B
        if not isinstance(model, (SPVCNN, MinkUNet)):
            raise TypeError('lidal_amd: Geometry.build knows the coordinate work of SPVCNN and MinkUNet, not of %s'
                            % type(model).__name__)
        B.require_gpu(coords)
        if coords.dtype != torch.int32 or coords.dim() != 2 or coords.shape[1] != 4 or not coords.is_contiguous():
            raise ValueError('lidal_amd: Geometry.build wants contiguous int32 coordinates [N, 4] = (x, y, z, batch), got '
                             '%s %s' % (coords.dtype, tuple(coords.shape)))
        if grad is None:
            grad = model.training and torch.is_grad_enabled()
        g = Geometry(coords, type(model).__name__, bool(grad))
        g._arena = arena if arena is not None else B.BlockArena()
        # (the map builder includes the rule lists iff gradients are on)
        with torch.set_grad_enabled(bool(grad)), B.use_arena(g._arena):
            if isinstance(model, SPVCNN):
                z = PointTensor(None, coords.float())
                x0 = SparseTensor(None, initial_tables(z, model.pres, model.vres), 1)
                x0.cmaps.setdefault(x0.stride, x0.coords)
                prefetch_kernel_maps(x0, model.MAP_PLAN)
                for s in model.POINT_STRIDES:
                    xs = SparseTensor(None, x0.cmaps[(s, s, s)], s)
                    xs.cmaps, xs.kmaps = x0.cmaps, x0.kmaps
                    idx, w = corner_tables(xs, z, own_cells=True)
                    pidx, counts = point_tables(xs, z)
                    prepare_voxelize(pidx, counts)
                    idx._lidal_cell_index = _index32(pidx)      # (one list of the points per voxel for both directions)
                    if grad:
                        # (channels of the features devoxelized at this stride, network/spvcnn.py:139-155: the
                        #  backward's lists are per-voxel or per-cell by them -- F.devoxelize.cells_mode)
                        pc = {1: model.cs[8], 16: model.cs[4], 4: model.cs[6]}.get(s)
                        prepare_devoxelize(idx, w, xs.C.shape[0], pc)
                g.z = z
            else:
                x0 = SparseTensor(None, coords, 1)
                prefetch_kernel_maps(x0, model.MAP_PLAN)
            g.x0 = x0
        import weakref
        me = weakref.ref(g)
        for km in x0.kmaps.values():        # a backward pass meets the geometry through its kernel maps (conv_backward)
            km._owner = me
        return g

    def admit(self, x, kind):
        """The checks of enter() without the feature work: the tables are x's, they are still alive, and the current
        stream waits for the stream that built them.  (What a planned step calls, network/plan.py.)"""
        if kind != self.kind:
            raise RuntimeError('lidal_amd: geometry built for %s handed to %s' % (self.kind, kind))
        c = x.C
        if tuple(x.s) != (1, 1, 1):
            raise RuntimeError('lidal_amd: geometries are built for stride-1 inputs, got stride %s' % (x.s,))
        if not (c is self.coords or (c.data_ptr() == self.coords.data_ptr() and c.shape == self.coords.shape
                                     and c.dtype == self.coords.dtype)):
            raise RuntimeError('lidal_amd: x.geometry was built for other coordinates than x.C')
        if self.ready is not None:
            if self._age >= 2:
                raise RuntimeError('lidal_amd: this geometry is stale -- two newer ones have been submitted to its '
                                   'prefetcher since; its memory may already serve another build (GeometryPrefetcher)')
            cur = torch.cuda.current_stream(c.device)
            if cur != self._stream:
                cur.wait_event(self.ready)
            self._consumer = cur                # the prefetcher fences THIS stream before it lets the tables go

    def alive(self):
        """May a kernel queued NOW still read the tables?  (False once two newer geometries have been submitted to the
        prefetcher that built this one: the fence that protects the memory has been recorded by then.)"""
        return self.ready is None or self._age < 2

    def enter(self, x, kind):
        """Called by the model's forward with its input: checks that the tables are x's, makes the current stream
        wait for them, and returns (x0 with x's features, z or None)."""
        self.admit(x, kind)
        if self.z is None:
            x0 = SparseTensor(x.F, x.C, x.s)
            x0.cmaps, x0.kmaps = self.x0.cmaps, self.x0.kmaps
            return x0, None
        z = PointTensor(x.F, self.z.C, idx_query=self.z.idx_query, weights=self.z.weights)
        z.additional_features = self.z.additional_features
        feats = F.spvoxelize(z.F, z.additional_features['idx_query'][1], z.additional_features['counts'][1])
        x0 = SparseTensor(feats, self.x0.C, 1)
        x0.cmaps, x0.kmaps = self.x0.cmaps, self.x0.kmaps
        return x0, z


# per device: the second stream (module state, not prefetcher state: the caching allocator keeps one pool per stream, and
# a new stream per prefetcher would start from an empty pool every time) and the geometries of prefetchers that went
# away while a consumer stream could still be reading their tables.
_STATE = {}


def _state(device):
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _STATE:
        _STATE[key] = {'stream': torch.cuda.Stream(device=device, priority=-1), 'orphans': []}
    return _STATE[key]


def _reap(st):
    st['orphans'] = [h for h in st['orphans'] if not h[1].query()]


class GeometryPrefetcher:
    """Builds geometries on a second stream of `device`:

        g = pf.submit(coords_0)
        for batch in batches:
            train_step(model, opt, feats, coords, labels, geometry=g)     # queues the step on the current stream
            g = pf.submit(next_coords)        # next batch's tables, built beside that step on the GPU

    submit() returns when the tables' sizes are known to the host (it waits for ITS stream only).  `coords` must
    be valid on submission (resident, or pass the event after which they are as `ready=`).

    Memory.  The tables are allocated on the second stream and read by the consumer's stream, so their memory
    must not return to the second stream's pool while a consumer kernel may still read it.  Instead of marking
    each of the ~150 tensors for the allocator (record_stream: an event per tensor and step, and -- measured --
    blocks that cannot be re-used in time, so the pool kept growing by hipMalloc), the prefetcher keeps every
    geometry alive itself: at the SECOND submit() of this prefetcher after a geometry's own it records one event on
    the stream that consumed the geometry, and lets go of the geometry once that event has passed.  The contract
    that makes this safe: whatever consumes a geometry -- forward AND backward pass -- is queued before the second
    submit() after its own (the loop above queues it before the first); a geometry handed to a forward pass, or
    met by a backward pass, later than that is refused (RuntimeError), not raced.  The consumer of a geometry is
    the stream on which its forward pass runs (until then: the stream that was current at its submit()).  Ages and
    fences are per prefetcher: another prefetcher on the same device (the scorer's, inside a training loop) neither
    ages this one's geometries nor fences them on its own stream."""

    MAX_PENDING = 1

    def __init__(self, model, device=None):
        self.model = model
        if device is None:
            device = next(model.parameters()).device
        self.device = torch.device(device)
        self._st = _state(self.device)
        self.stream = self._st['stream']
        self.held = []              # [geometry, fence event or None]

    def submit(self, coords, grad=None, ready=None, arena=None):
        if grad is None:
            grad = self.model.training and torch.is_grad_enabled()
        fences = {}
        for h in self.held:
            g = h[0]
            g._age += 1
            if g._age == 2 and h[1] is None:
                c = g._consumer
                if c not in fences:
                    fences[c] = c.record_event()
                h[1] = fences[c]
        # bounded run-ahead: at most MAX_PENDING fenced generations may still be waiting for their consumer -- a host
        # that queues steps faster than the GPU runs them (nothing else synchronises the loop) would otherwise keep one
        # more generation of tables alive per step of lead
        pending = [h for h in self.held if h[1] is not None and not h[1].query()]
        while len(pending) > self.MAX_PENDING:
            pending.pop(0)[1].synchronize()
        self.held = [h for h in self.held if h[1] is None or not h[1].query()]
        _reap(self._st)
        consumer = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            if ready is not None:
                self.stream.wait_event(ready)
            coords.record_stream(self.stream)
            g = Geometry.build(self.model, coords, grad, arena)
            g.ready = self.stream.record_event()
            g._stream = self.stream
            g._consumer = consumer
        self.held.append([g, None])
        return g

    def submit_batch(self, make, grad=None):
        """The INPUT of a step built ahead as well: `make()` is called under the second stream and returns the batch
        as a dict with 'coords_v_b' (what lidal_amd.data.collate returns: the scans of the step voxelised under a
        newly drawn augmentation, dataset/sk_dataset.py:143-171 + :188-242 -- the work the reference's DataLoader
        workers do ahead of the GPU, dataset/sk_dataloader.py:21,53); its coordinate tables follow on the same
        stream.  Returns the geometry with the batch as `.payload`: the batch tensors were allocated on the second
        stream and live exactly as long as the tables do (the same fence)."""
        if grad is None:
            grad = self.model.training and torch.is_grad_enabled()
        from .. import backend as B
        arena = B.BlockArena()                  # the voxeliser's buffers and the tables: one set of blocks, one lifetime
        with torch.cuda.stream(self.stream), B.use_arena(arena):
            batch = make()
        g = self.submit(batch['coords_v_b'], grad, arena=arena)
        g.payload = batch
        return g

    def drain(self):
        """Let go of every geometry this prefetcher still holds (the last two of a loop stay alive until the next
        submit): waits for their consumer streams, after which nothing can still be reading them."""
        for h in self.held:
            h[0]._consumer.synchronize()
            h[0]._age = max(h[0]._age, 2)           # a drained geometry must not be handed to a forward pass any more
        self.stream.synchronize()
        self.held = []
        _reap(self._st)

    def close(self):
        """The same without waiting on the host: every held geometry is fenced on its consumer stream now and kept by the
        device's module state until that event has passed.  Called when a loop ends (and on garbage collection)."""
        for h in self.held:
            g = h[0]
            g._age = max(g._age, 2)
            if h[1] is None:
                h[1] = g._consumer.record_event()
            self._st['orphans'].append(h)
        self.held = []
        _reap(self._st)

    def __del__(self):
        try:
            self.close()
        except Exception:           # noqa: BLE001  (interpreter shutdown: the runtime may already be gone)
            pass
