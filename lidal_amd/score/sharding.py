"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

  frame_range()     the reference's contiguous split, dataset/sk_dataloader.py:196-198:
                    rank r owns frames [r*ceil(F/G), (r+1)*ceil(F/G)).
  gather_frames()   replaces the reference's disk hand-off between score/prob_inference.py:129
                    (np.save per frame) and score/sv_level/LiDAL.py:45-49 (np.load of 25 frames):
                    ONE all_gather_into_tensor of a flat padded buffer of the per-frame [P, C]
                    probabilities (or world coords), so every rank holds every frame for the
                    +-nei window.
                    The same call carries the return leg (score/pipeline.py collect_sequence:
                    ~40 kB of per-supervoxel results per sequence, consumed by rank 0).
"""
import math

import torch
import torch.distributed as dist

__all__ = ['frame_range', 'gather_frames', 'collective_device', 'is_sharded']


def frame_range(n_frames, world_size, rank):
    per = math.ceil(n_frames / world_size)
    return range(min(rank * per, n_frames), min((rank + 1) * per, n_frames))


def is_sharded(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def collective_device(group=None):
    """Device the collectives of `group` need their tensors on: the current GPU under nccl (RCCL),
    the host under gloo.  Never derived from a local frame: a rank that owns no frame (9 frames on
    8 GPUs) must still pass a tensor of the right kind."""
    if dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def gather_frames(local, n_frames, tail, dtype, group=None, device=None):
    """local: dict frame_id -> tensor [P_f, *tail] of `dtype` for the frames this rank owns (may
    be empty).  Returns a list of n_frames tensors, identical on every rank, on `device` (default:
    where the local frames live, or the collective's device for a rank without frames)."""
    tail = tuple(tail)
    if not is_sharded(group):
        return [local[f] for f in range(n_frames)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = math.ceil(n_frames / world)
    mine = list(frame_range(n_frames, world, rank))
    cdev = collective_device(group)
    out_dev = device or (local[mine[0]].device if mine else cdev)
    # 1. lengths of every frame (one tiny all-gather)
    lens = torch.zeros(per, dtype=torch.int64)
    for s, f in enumerate(mine):
        assert tuple(local[f].shape[1:]) == tail and local[f].dtype == dtype
        lens[s] = local[f].shape[0]
    lens = lens.to(cdev)
    all_lens = torch.empty(world * per, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(all_lens, lens, group=group)
    all_lens = all_lens.cpu().view(world, per)
    pmax = int(all_lens.max().item())
    # 2. one flat padded block per rank, one collective
    block = torch.zeros((per, pmax) + tail, dtype=dtype, device=cdev)
    for s, f in enumerate(mine):
        block[s, :local[f].shape[0]] = local[f].to(cdev)
    blocks = torch.empty((world, per, pmax) + tail, dtype=dtype, device=cdev)
    dist.all_gather_into_tensor(blocks.view(-1), block.view(-1), group=group)
    if blocks.device != out_dev:
        blocks = blocks.to(out_dev)
    out = []
    for f in range(n_frames):
        r, s = divmod(f, per)
        out.append(blocks[r, s, :int(all_lens[r, s])])
    return out
