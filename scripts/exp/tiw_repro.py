"""Minimal form of the geometry_stress.py finding: lidal_ti_weights on a side stream while the main stream is busy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd.nn import functional as F

dev = torch.device('cuda')
torch.manual_seed(0)
n = int(os.environ.get('N', '396662'))
coords = (torch.rand(n, 4, device=dev) * 4000).floor()
coords[:, 0] += (torch.rand(n, device=dev) < 0.3).float() * 0.000488        # as the reference's float noise
idx = torch.randint(0, n, (8, n), device=dev, dtype=torch.int64)
idx[torch.rand(8, n, device=dev) < 0.5] = -1
side = torch.cuda.Stream()
busy = os.environ.get('BUSY', 'matmul')
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
for scale in (1, 4, 16):
    ref_w, ref_i = F.ti_weights_and_index(coords, idx, scale)
    torch.cuda.synchronize()
    bad_alone = bad_beside = 0
    for it in range(100):
        w, i32 = F.ti_weights_and_index(coords, idx, scale)
        torch.cuda.synchronize()
        bad_alone += int(not torch.equal(w, ref_w))
    rows = set()
    for it in range(100):
        if busy == 'matmul':
            for _ in range(4):
                a @ a
        elif busy == 'fill':
            for _ in range(8):
                big.fill_(it & 255)
        with torch.cuda.stream(side):
            w, i32 = F.ti_weights_and_index(coords, idx, scale)
        torch.cuda.synchronize()
        if not torch.equal(w, ref_w):
            bad_beside += 1
            d = torch.nonzero((w != ref_w))
            rows.update(d[:, 1].tolist())
    print('scale %2d: differs from the first run in %d / 100 runs alone, %d / 100 beside %s; columns that differ: %s'
          % (scale, bad_alone, bad_beside, busy, sorted(rows)))
