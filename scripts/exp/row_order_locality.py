"""EXPERIMENT (round 4): how much of the level-0 convolution's line traffic (2.9 x algorithmic, profiles/r04_pmc_conv_
apply.json) is the ROW ORDER of its tiles?  lidal_kmap_order sorts the rows of a table by occupancy pattern over the
WHOLE level (7 active offsets per 128-row tile instead of 15-20), which scatters every tile's rows over the scene --
no two gathers of a tile share a cache line, whatever the voxel numbering.  The convolution's result does not depend
on the order (a row's offsets are summed in offset order), so every variant below is built here in torch and handed to
the shipped kernel:

  shipped         lidal_kmap_order (pattern sort over the level)
  block B         rows in blocks of B consecutive voxels (Z-order numbering: a block is a compact blob), pattern
                  sort inside a block only
  identity        B = 1 tile: pure Z-order tiles
  +xcd            tiles dealt so that workgroup p (XCD p % 8) takes logical tile (p % 8) * ceil(T / 8) + p // 8:
                  each XCD's L2 walks one contiguous eighth of the level

usage: LIDAL_L0_ORDER=morton|hash python scripts/exp/row_order_locality.py        (REPS, VARIANT=name to run one)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lidal_amd import backend as B, synth  # noqa: E402
if os.environ.get('LIDAL_L0_ORDER', 'morton') == 'morton':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import l0_morton  # noqa: E402,F401
from lidal_amd.network import SPVCNN, Geometry  # noqa: E402
from lidal_amd.nn.functional.conv import RowOrder, _weight_image  # noqa: E402

dev = torch.device('cuda', 0)
FRAMES = int(os.environ.get('FRAMES', '5'))
b = synth.make_train_batch(n_frames=FRAMES, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).train()
g = Geometry.build(model, coords, True)
km = g.x0.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n = km.sizes[0]
nbr = km.nbr_out                           # i32 [27, n]
K = 27


def pattern_keys(nbr):
    """The sort key of lidal_kmap_order (csrc/kmap.hip: bit_rank + Gray rank) and the plain masks."""
    to_key = list(range(32))
    pos = 26
    for want in (3, 2, 1, 0):
        for k in range(27):
            a, bb, c = k % 3, (k // 3) % 3, k // 9
            if (a != 1) + (bb != 1) + (c != 1) == want:
                to_key[k] = pos
                pos -= 1
    m = torch.zeros(nbr.shape[1], dtype=torch.int64, device=nbr.device)
    plain = torch.zeros_like(m)
    for k in range(27):
        occ = (nbr[k] >= 0).long()
        m |= occ << to_key[k]
        plain |= occ << k
    for s in (1, 2, 4, 8, 16):
        m ^= m >> s
    return m, plain


KEYS, PLAIN = pattern_keys(nbr)


def make_order(perm):
    o = RowOrder(nbr, build=False)
    o.perm = perm.int().contiguous()
    o.table = nbr[:, perm].contiguous()
    tiles = -(-n // 128)
    pm = torch.zeros(tiles * 128, dtype=torch.int64, device=dev)
    pm[:n] = PLAIN[perm]
    t = pm.view(tiles, 128)
    acc = t[:, 0].clone()
    for j in range(1, 128):
        acc |= t[:, j]
    o.tile_masks = acc.int().contiguous()
    return o


def xcd_deal(perm):
    """Whole tiles re-dealt: dispatched tile p <- logical tile (p % 8) * ceil(T/8) + p // 8 (the last, ragged tile stays)."""
    full = n // 128
    per = -(-full // 8)
    p = torch.arange(full, device=dev)
    logical = (p % 8) * per + p // 8
    # logical tiles >= full do not exist: compact the sequence, keeping the dealing order
    logical = logical[logical < full]
    rest = torch.tensor(sorted(set(range(full)) - set(logical.tolist())), dtype=torch.long, device=dev)
    logical = torch.cat([logical, rest])
    rows = (logical[:, None] * 128 + torch.arange(128, device=dev)[None, :]).reshape(-1)
    return torch.cat([perm[rows], perm[full * 128:]])


def block_perm(block):
    rows = torch.arange(n, device=dev)
    key = ((rows // block) << 27) | KEYS
    return torch.argsort(key, stable=True)


variants = {'shipped': None, 'shipped(rebuilt)': km.order_out.perm.long()}
for blk in (128, 1024, 4096, 16384, 65536):
    variants['block%d' % blk] = block_perm(blk)
variants['pattern(torch)'] = torch.argsort(KEYS, stable=True)
for name in list(variants):
    if variants[name] is not None:
        variants[name + '+xcd'] = xcd_deal(variants[name])
only = os.environ.get('VARIANT')
x = torch.randn(n, 96, device=dev).bfloat16()
img = _weight_image(torch.randn(27, 96, 96, device=dev) * 0.02, torch.bfloat16, n, 0)
L = B.lib()
reps = int(os.environ.get('REPS', '10'))
ref = None
print(os.environ.get('LIDAL_L0_ORDER', 'morton'), 'rows', n, 'rules', int((nbr >= 0).sum()))
for name, perm in variants.items():
    if only and name != only:
        continue
    order = km.order_out if perm is None else make_order(perm)
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
    print('%-22s %7.1f us   active offsets per tile %5.2f   bitwise %s' % (name, ev[0].elapsed_time(ev[1]) * 1e3 / reps, active,
                                                                          torch.equal(out, ref)))
