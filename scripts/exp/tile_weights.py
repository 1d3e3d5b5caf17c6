"""Phases per tile along the dispatch order of the level-0 3x3x3 map (5-scan bench batch): where do the heavy tiles sit?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import synth
from lidal_amd.nn import functional as F
batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords = torch.from_numpy(batch['coords_v_b']).cuda()
kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
m = kmap.order_out.tile_masks.cpu().numpy().astype(np.uint32)
pc = np.array([bin(int(v)).count('1') for v in m])
print('tiles', len(pc), 'mean phases', pc.mean(), 'max', pc.max())
for i, part in enumerate(np.array_split(pc, 12)):
    print('  twelfth %2d of the order: mean %.2f  max %d' % (i, part.mean(), part.max()))
