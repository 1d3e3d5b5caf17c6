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
  * every backward pass: a post-accumulate hook on the parameters queues ONE end-of-backward callback; it scales by
    1 / world and all-reduces (sum) -- in place as ONE tensor when the gradients lie back to back in one storage (the
    planned step), else through a flattened copy (the per-operator path, any other module);
  * BatchNorm running statistics stay per rank (DDP's `broadcast_buffers=False`): rank 0's, the ones a checkpoint
    holds (train.py:150-155 saves on rank 0), are the statistics DDP's per-forward broadcast would keep as well, since
    that broadcast only ever overwrites the OTHER ranks'.

Same gradients as DDP (mean over ranks, scaled before the sum); `state_dict()` carries the 'module.' prefix like DDP's
(lidal_amd.io strips it)."""
import torch
import torch.distributed as dist

__all__ = ['DataParallel']


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
        self._params = [p for p in module.parameters() if p.requires_grad]
        self.reductions = 0             # backward passes reduced
        self.flat_reductions = 0        # ... of which in place on one flat buffer
        if broadcast:
            self._broadcast_state()
        for p in self._params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

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
    def _on_grad(self, param):
        if not self._queued:
            self._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce)

    @staticmethod
    def _back_to_back(grads):
        """The gradients as ONE tensor if they are contiguous f32 views lying back to back (gaps of less than 16 bytes:
        16-byte slots) in one storage -- the planned step's flat buffer -- else None."""
        g0 = grads[0]
        if any(g.dtype != torch.float32 or not g.is_contiguous() or g.device != g0.device for g in grads):
            return None
        st = g0.untyped_storage()
        if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
            return None
        spans = sorted((g.storage_offset(), g.numel()) for g in grads)
        for (a, n), (b, _) in zip(spans, spans[1:]):
            if not 0 <= b - (a + n) < 4:
                return None
        lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, lo, (hi - lo,))

    def _reduce(self):
        self._queued = False
        grads = [p.grad for p in self._params if p.grad is not None]
        if not grads:
            return
        self.reductions += 1
        flat = self._back_to_back(grads)
        if flat is not None:
            self.flat_reductions += 1
            flat.mul_(1.0 / self.world)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
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
