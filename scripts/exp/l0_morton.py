"""EXPERIMENT (round 4, verdict item 5): SPVCNN's level-0 voxels renumbered along a Z-order curve instead of by
sorted coordinate hash (the reference's torch.unique order, network/utils.py:18; the order is internal -- logits
return per point).  `import l0_morton` (or LIDAL_L0_ORDER=morton with the scripts of this directory) installs the hook.

Measured on MI355X (profiles/r04_l0_order_*): the level-0 96->96 convolution 93.3 -> 87.4 us (line traffic 520 -> 451 MB),
its weight gradient 123.5 -> 116.1 us (861 -> 801 MB); over the whole 5-scan step conv_apply 6.27 -> 6.27 ms, conv_wgrad
3.05 -> 3.01 ms.  The gate (weight gradient <= 85 us or <= 600 MB) is far off: the kernels are not bound by the locality
of their gathers.  Not adopted."""
import os

import torch

from lidal_amd.network import glue

_MORTON_SHIFT = int(os.environ.get('LIDAL_MORTON_SHIFT', '0'))


def _part1by2(v):
    v = v & 0x1FFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def morton_renumber(idx_query, counts, coords):
    """Level-0 voxel rows re-ordered by (batch, Morton(x, y, z)) (torch ops: an experiment)."""
    c = coords.long()
    key = (c[:, 3] << 48) | _part1by2(c[:, 0] >> _MORTON_SHIFT) | (_part1by2(c[:, 1] >> _MORTON_SHIFT) << 1) \
        | (_part1by2(c[:, 2] >> _MORTON_SHIFT) << 2)
    perm = torch.argsort(key, stable=True)              # sorted position -> old row
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    new_idx = inv[idx_query]
    if getattr(idx_query, '_lidal_one_to_one', False):
        new_idx._lidal_one_to_one = True
    return new_idx, counts[perm].contiguous(), coords[perm].contiguous()


if os.environ.get('LIDAL_L0_ORDER', 'morton') == 'morton':
    glue.RENUMBER = morton_renumber
