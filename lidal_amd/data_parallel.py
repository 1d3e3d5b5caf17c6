"""Data parallelism for the training step: the counterpart of /root/reference/train.py:49-53
(`torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank], output_device=rank)`), one process per GPU,
gradients averaged over the ranks inside `loss.backward()`, RCCL underneath (`torch.distributed`, backend "nccl").

torch's DistributedDataParallel works with this package (tests/test_multirank_gpu.py) but is built for a graph of many
autograd nodes whose gradients arrive one by one: per step it copies every gradient into a bucket and back (2 x 161
launches for SPVCNN), reduces bucket by bucket and broadcasts ~150 BatchNorm buffers before every forward pass.  Measured
on one MI355X with one rank over RCCL: the 5-scan step 15.2 -> 17.2 ms, the single-scan step 6.7 -> 12.7 ms, before any
byte has crossed xGMI.  The planned step (network/plan.py) is ONE autograd node whose parameter gradients are views of
one flat f32 buffer, complete at the same moment -- so here the reduction is one collective at the end of the backward
pass:

  * construction: parameters and buffers of rank 0 are broadcast once (as DDP does);
  * every backward pass: a hook on the forward pass's outputs queues ONE end-of-backward callback when the backward pass
    reaches them (one Python call per step; a post-accumulate hook on each of the 161 parameters cost 0.3 ms of host
    time); the callback scales by 1 / world and all-reduces (sum) -- in place as ONE tensor when the gradients lie back to
    back in one storage (the planned step), else through a flattened copy (the per-operator path, any other module).
    Three sentinel parameters tell it that this backward pass accumulated parameter gradients at all (a
    `torch.autograd.grad` call through the outputs does not);
  * BatchNorm running statistics stay per rank (DDP's `broadcast_buffers=False`): rank 0's, the ones a checkpoint
    holds (train.py:150-155 saves on rank 0), are the statistics DDP's per-forward broadcast would keep as well, since
    that broadcast only ever overwrites the OTHER ranks'.

Same gradients as DDP (mean over ranks, scaled before the sum); `state_dict()` carries the 'module.' prefix like DDP's
(lidal_amd.io strips it)."""
import torch
import torch.distributed as dist

__all__ = ['DataParallel']

import os as _os
# one rank: run the (trivial) collective anyway -- how bench.py's BENCH_FORCE_DDP=1 executes the RCCL path on a 1-GPU box
_FORCE = _os.environ.get('LIDAL_DP_FORCE_COLLECTIVE', '0') != '0'


class DataParallel(torch.nn.Module):
    def __init__(self, module, process_group=None, broadcast=True):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('lidal_amd.DataParallel needs an initialised torch.distributed process group '
                               '(train.py:27-28: dist.init_process_group)')
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._queued = False
        self._touched = False
        self._layout = None             # (address offsets of the gradients, index of the lowest, elements) once they lay back to back
        self._params = [p for p in module.parameters() if p.requires_grad]
        self.reductions = 0             # backward passes reduced
        self.flat_reductions = 0        # ... of which in place on one flat buffer
        # The two forms of the reduction must put the SAME number of elements in the SAME order on the wire, because
        # each rank picks its form locally (a rank whose step was not plannable -- a batch of <= 1 rows, plan.ENABLED
        # off, a pre-existing .grad -- reduces through a copy while the others reduce in place).  So for this package's
        # networks the copy is laid out exactly like the planned step's flat buffer (network/plan.py: slot offsets,
        # 16-byte gaps, classifier bias last): compiled here, on every rank alike, whether or not plans are enabled.
        self._prog = None
        try:
            from .network import plan as _plan
            prog = _plan._program(module, module.training)
            if prog is not None and len(prog.params) == len(self._params) and \
                    all(a is b for a, b in zip(sorted(prog.params, key=id), sorted(self._params, key=id))):
                self._prog = prog
        except Exception:               # noqa: BLE001  (any other module: parameters() order on every rank)
            self._prog = None
        if broadcast:
            self._broadcast_state()
        n = len(self._params)
        for i in sorted({0, n // 2, n - 1} if n else ()):
            self._params[i].register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled():
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    t.register_hook(self._arm)
        return out

    # ---- once: rank 0's parameters and buffers to every rank ------------------------------------------------
    def _broadcast_state(self):
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        by_type = {}
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            by_type.setdefault((t.dtype, t.device), []).append(t.data)
        for tensors in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in tensors])
            dist.broadcast(flat, src, group=self.group)
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    # ---- every backward pass ---------------------------------------------------------------------------------
    def _arm(self, grad):
        """The backward pass has reached an output of this module's forward pass."""
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce)
        return None

    def _on_grad(self, param):
        self._touched = True
        self._arm(None)                 # (a module whose outputs are not tensors / tuples of tensors)

    def _back_to_back(self, grads):
        """The gradients as ONE tensor if they are contiguous f32 views lying back to back (gaps of less than 16 bytes:
        16-byte slots) in one storage -- the planned step's flat buffer -- else None.  The full check (dtype, layout,
        storage, order) runs once; afterwards a step only compares the 161 addresses with the layout found then
        (25 us instead of 1 ms of Python on a host-bound step)."""
        g0 = grads[0]
        lay = self._layout
        if lay is not None and len(grads) == len(lay[0]):
            rel, lo_index, span = lay
            base = grads[lo_index].data_ptr()
            st = grads[lo_index].untyped_storage()
            if ([g.data_ptr() - base for g in grads] == rel and 0 <= base - st.data_ptr()
                    and base - st.data_ptr() + 4 * span <= st.nbytes()):
                return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, (base - st.data_ptr()) // 4, (span,))
        self._layout = None
        if any(g.dtype != torch.float32 or not g.is_contiguous() or g.device != g0.device for g in grads):
            return None
        st = g0.untyped_storage()
        if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
            return None
        spans = sorted((g.storage_offset(), g.numel(), i) for i, g in enumerate(grads))
        for (a, n, _), (b, _, _) in zip(spans, spans[1:]):
            if not 0 <= b - (a + n) < 4:
                return None
        lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
        base = grads[spans[0][2]].data_ptr()
        self._layout = ([g.data_ptr() - base for g in grads], spans[0][2], hi - lo)
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, lo, (hi - lo,))

    def _reduce(self):
        self._queued = False
        touched, self._touched = self._touched, False
        grads = [p.grad for p in self._params if p.grad is not None]
        if not grads or not touched:
            return
        self.reductions += 1
        if self.world == 1 and not _FORCE:      # (nothing to average; LIDAL_DP_FORCE_COLLECTIVE=1 runs the collective anyway)
            return
        flat = self._back_to_back(grads)
        if flat is not None:
            self.flat_reductions += 1
            flat.mul_(1.0 / self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        prog = self._prog
        if prog is not None and len(grads) == len(prog.params) and all(p.grad is not None for p in prog.params):
            # the planned step's layout, through a copy: what the in-place form of the other ranks sends
            buf = torch.zeros(prog.flat_numel, dtype=torch.float32, device=grads[0].device)
            views = [buf[prog.slot[i]:prog.slot[i] + p.numel()].view_as(p) for i, p in enumerate(prog.params)]
            gs = [p.grad for p in prog.params]
            torch._foreach_copy_(views, gs)
            buf.mul_(1.0 / self.world)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            torch._foreach_copy_(gs, views)
            return
        by_type = {}
        for g in grads:
            by_type.setdefault((g.dtype, g.device), []).append(g)
        for tensors in by_type.values():
            flat = torch.cat([g.reshape(-1) for g in tensors])
            flat.mul_(1.0 / self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for g in tensors:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
