"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

  frame_range()    the reference's contiguous split, dataset/sk_dataloader.py:196-198:
                   rank r owns frames [r*ceil(F/G), (r+1)*ceil(F/G)).
  gather_frames()  replaces the reference's disk hand-off between score/prob_inference.py:129
                   (np.save per frame) and score/sv_level/LiDAL.py:45-49 (np.load of 25 frames):
                   ONE all-gather of the per-frame [P, C] probabilities (and world coords), padded
                   to the longest frame, so every rank holds every frame for the +-nei window.
"""
import math

import torch
import torch.distributed as dist

__all__ = ['frame_range', 'gather_frames']


def frame_range(n_frames, world_size, rank):
    per = math.ceil(n_frames / world_size)
    return range(min(rank * per, n_frames), min((rank + 1) * per, n_frames))


def gather_frames(local, n_frames, group=None):
    """local: dict frame_id -> tensor [P_f, ...] for the frames this rank owns (same trailing
    shape and dtype everywhere).  Returns a list of n_frames tensors, identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local[f] for f in range(n_frames)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = math.ceil(n_frames / world)
    mine = list(frame_range(n_frames, world, rank))
    ref = local[mine[0]] if mine else None
    # 1. lengths of every frame (tiny all-gather)
    dev = ref.device if ref is not None else torch.device('cpu')
    lens = torch.zeros(per, dtype=torch.int64, device=dev)
    for s, f in enumerate(mine):
        lens[s] = local[f].shape[0]
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    all_lens = torch.stack(all_lens).cpu()
    pmax = int(all_lens.max().item())
    # trailing shape / dtype must be agreed on even by ranks without frames
    meta = [None] * world
    dist.all_gather_object(meta, None if ref is None else (tuple(ref.shape[1:]), str(ref.dtype)),
                           group=group)
    tail, dt = next(m for m in meta if m is not None)
    dtype = getattr(torch, dt.replace('torch.', ''))
    # 2. one padded block per rank
    block = torch.zeros((per, pmax) + tuple(tail), dtype=dtype, device=dev)
    for s, f in enumerate(mine):
        block[s, :local[f].shape[0]] = local[f]
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    out = []
    for f in range(n_frames):
        r, s = divmod(f, per)
        out.append(blocks[r][s, :int(all_lens[r, s])])
    return out
