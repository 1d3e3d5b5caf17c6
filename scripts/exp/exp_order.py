"""GPU experiment: row-order variants for lidal_conv_apply WITHOUT kernel changes -- the order is
data (perm, permuted table, per-tile masks).  Variants: global Gray-rank mask sort (shipped),
mask sort within blocks of B rows of the memory order, each with and without an XCD-aware tile
placement (workgroup b runs on XCD b % 8: logical tile x*T/8 + i is stored at physical slot 8*i + x,
so every XCD walks a contiguous range of tiles and neighbouring tiles share its L2).
  python scripts/exp_order.py [ci co] ...
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402


def gray_keys(nbr):
    rank = np.zeros(27, np.int64)
    pos = 26
    for want in (3, 2, 1, 0):
        for k in range(27):
            a, b, c = k % 3, (k // 3) % 3, k // 9
            if (a != 1) + (b != 1) + (c != 1) == want:
                rank[k] = pos
                pos -= 1
    m = np.zeros(nbr.shape[1], np.int64)
    for k in range(27):
        m |= (nbr[k] >= 0).astype(np.int64) << rank[k]
    g = m.copy()
    for s in (1, 2, 4, 8, 16):
        g ^= g >> s
    return g


def tables(nbr, perm, xcd):
    """perm: sorted position -> row.  Returns (perm, table, tile_masks) as the kernel wants them,
    tiles optionally re-placed for XCD locality (whole 128-row tiles move; the last partial tile stays last)."""
    n = nbr.shape[1]
    tiles = (n + 127) // 128
    if xcd:
        full = n // 128                       # only full tiles are moved
        per = full // 8
        phys = np.arange(tiles)
        logical = np.arange(per * 8).reshape(8, per).T.reshape(-1)    # physical slot 8*i+x <- logical x*per+i
        phys[:per * 8] = logical
        rows = (phys[:, None] * 128 + np.arange(128)[None, :]).reshape(-1)
        rows = rows[rows < n]
        perm = perm[rows]
    tab = nbr[:, perm]
    pad = tiles * 128 - n
    occ = np.concatenate([tab >= 0, np.zeros((27, pad), bool)], 1).reshape(27, tiles, 128).any(2)
    masks = (occ.astype(np.int64) << np.arange(27)[:, None]).sum(0).astype(np.uint32).view(np.int32)
    return perm.astype(np.int32), np.ascontiguousarray(tab.astype(np.int32)), masks, float(occ.sum(0).mean())


def main():
    shapes = [(96, 96)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    dev = torch.device('cuda')
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    order_mode = os.environ.get('EXP_MEMORY_ORDER', 'dataset')
    coords_np = batch['coords_v_b']
    if order_mode == 'hash':
        coords_np = coords_np[np.random.default_rng(0).permutation(len(coords_np))]
    coords = torch.from_numpy(coords_np).to(dev)
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    nbr = kmap.nbr_out.cpu().numpy()
    n = nbr.shape[1]
    g = gray_keys(nbr)
    variants = {'global': np.argsort(g, kind='stable')}
    for blk in (2048, 4096, 8192, 16384, 32768):
        variants['block%d' % blk] = np.lexsort((g, np.arange(n) // blk))
    variants['unsorted'] = np.arange(n)
    print('memory order: %s, rows %d, rules %d' % (order_mode, n, int((nbr >= 0).sum())))
    for ci, co in shapes:
        dtype = torch.bfloat16
        x = torch.randn(n, ci, device=dev).to(dtype)
        wk = (torch.randn(27, co, ci, device=dev) * 0.02).to(dtype)
        ref = None
        print('--- %d -> %d' % (ci, co))
        for name, perm in variants.items():
            for xcd in (False, True):
                p, t, m, act = tables(nbr, perm, xcd)
                pd, td, md = (torch.from_numpy(a).to(dev) for a in (p, t, m))
                out = torch.empty((n, co), dtype=dtype, device=dev)

                def launch():
                    B.check(B.lib().lidal_conv_apply(B.ptr(x), B.ptr(wk), B.ptr(td), B.ptr(pd), B.ptr(md),
                                                     B.ptr(out), n, n, ci, co, 27, 0, B.dtype_code(dtype),
                                                     None, None, 0, None, B.stream()), 'conv')
                for _ in range(3):
                    launch()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                same = torch.equal(out, ref)
                ts = []
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        launch()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 100)
                print('%-12s xcd=%d  act/tile %5.2f  %7.1f us (min of 3: %s)  bit-equal %s'
                      % (name, xcd, act, min(ts), ' '.join('%.1f' % t for t in ts), same), flush=True)


if __name__ == '__main__':
    main()
