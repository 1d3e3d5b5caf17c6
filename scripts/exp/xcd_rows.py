"""EXPERIMENT (round 6): the level-0 convolution with its tiles made PER XCD.  Rows get the spatial blocks of the
streamed weight gradient (csrc/wgrad_streams.hip: ~BLOCK consecutive parent rows, block b -> XCD b % 8); the pattern sort
of lidal_kmap_order runs inside each XCD's eighth of the rows (spread over the whole scene, so the patterns stay as
varied as in blocks of 50 k rows), and tile t is placed where the hardware will run it on that XCD (workgroup
T - 1 - t, XCD = workgroup % 8).  A tile's gathers then fall into its XCD's eighth of the input (~12.7 MB of lines
against a 4 MB L2: ~30 % hits instead of ~5 %) at the price of more active offsets per tile.  Orders built in torch,
handed to the shipped kernel (bitwise the same result), as scripts/exp/row_order_locality.py does.
usage: BLOCK=1024 REPS=20 python scripts/exp/xcd_rows.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('LIDAL_L0_ORDER', 'hash')
import torch  # noqa: E402

import row_order_locality as R  # noqa: E402  (builds the geometry, the pattern keys and make_order; runs its own table first)
from lidal_amd import backend as B  # noqa: E402

dev, n, km, nbr, KEYS = R.dev, R.n, R.km, R.nbr, R.KEYS
g = R.g
k2 = g.x0.kmaps[((1, 1, 1), (2, 2, 2), (2, 2, 2), (1, 1, 1))]
parent = k2.nbr_in.max(0)[0].long()
n1 = k2.sizes[1]


def xcd_order(block, key=None):
    nb = max(8, (n + 8 * block - 1) // (8 * block) * 8)
    kv = parent if key is None else key
    rng = n1 if key is None else n
    xcd = ((kv * nb) // rng) % 8
    tiles = -(-n // 128)
    lists = []
    for x in range(8):
        rows = torch.nonzero(xcd == x)[:, 0]
        lists.append(rows[torch.argsort(KEYS[rows], stable=True)])
    taken = [0] * 8
    out = []
    for t in range(tiles):
        x = (tiles - 1 - t) % 8
        want = min(128, n - 128 * t)
        got = []
        while want > 0:
            if taken[x] >= lists[x].numel():
                x = max(range(8), key=lambda q: lists[q].numel() - taken[q])
            m = min(want, lists[x].numel() - taken[x])
            got.append(lists[x][taken[x]:taken[x] + m])
            taken[x] += m
            want -= m
        out.append(torch.cat(got))
    return torch.cat(out)


variants = {'shipped': None}
for blk in (int(v) for v in os.environ.get('BLOCKS', '1024,4096').split(',')):
    variants['xcd eighths, blocks of %d rows' % blk] = xcd_order(blk)
variants['xcd eighths, random key (no locality)'] = xcd_order(1024, torch.randperm(n, device=dev))
x = torch.randn(n, 96, device=dev).bfloat16()
img = R._weight_image(torch.randn(27, 96, 96, device=dev) * 0.02, torch.bfloat16, n, 0)
L = B.lib()
reps = int(os.environ.get('REPS', '20'))
ref = None
print('--- per-XCD tiles')
for name, perm in variants.items():
    order = km.order_out if perm is None else R.make_order(perm)
    out = torch.empty((n, 96), dtype=torch.bfloat16, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(2):
        ev[0].record()
        for _ in range(reps):
            B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(order.table), B.ptr(order.perm), B.ptr(order.tile_masks),
                                             B.ptr(out), n, n, 96, 96, 27, 0, B.BF16, None, None, 0, None, None, B.stream()), 'conv')
        ev[1].record()
        torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    active = float(sum(((order.tile_masks.long() >> k) & 1).sum() for k in range(27))) / order.tile_masks.numel()
    print('%-44s %7.1f us   active offsets per tile %5.2f   bitwise %s' % (name, ev[0].elapsed_time(ev[1]) * 1e3 / reps, active,
                                                                          torch.equal(out, ref)))
