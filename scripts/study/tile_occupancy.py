"""CPU study: how many kernel offsets does a tile of R consecutive rows need after the occupancy
sort of lidal_kmap_order (rarest-offset-major key, Gray rank), per level of the bench batch, and
what fraction of its (row, offset) slots holds a rule.  The convolution walks a tile's active
offsets in lock-step (one weight slab per phase for all its waves), so slots without a rule are
MFMA and LDS work for nothing."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lidal_amd import synth  # noqa: E402


def key64(c):
    return ((c[:, 3].astype(np.int64) << 48) | ((c[:, 0].astype(np.int64) + 4096) << 32)
            | ((c[:, 1].astype(np.int64) + 4096) << 16) | (c[:, 2].astype(np.int64) + 4096))


def masks(coords, stride):
    ks = np.sort(key64(coords))
    m = np.zeros(len(coords), dtype=np.uint32)
    k = 0
    for dz in (-1, 0, 1):          # k = a + 3 b + 9 c, any fixed digit order does for statistics
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = coords.copy()
                q[:, 0] += dx * stride
                q[:, 1] += dy * stride
                q[:, 2] += dz * stride
                kk = key64(q)
                i = np.minimum(np.searchsorted(ks, kk), len(ks) - 1)
                m |= (ks[i] == kk).astype(np.uint32) << np.uint32(k)
                k += 1
    return m


def rank_key(m):
    order = []
    for want in (0, 1, 2, 3):      # least significant first: centre, faces, edges, corners
        for k in range(27):
            a, b, c = k % 3, (k // 3) % 3, k // 9
            if (a != 1) + (b != 1) + (c != 1) == want:
                order.append(k)
    key = np.zeros_like(m)
    for pos, k in enumerate(order):
        key |= ((m >> np.uint32(k)) & np.uint32(1)) << np.uint32(pos)
    g = key.copy()
    for s in (1, 2, 4, 8, 16):
        g ^= g >> np.uint32(s)
    return g


def popcount(x):
    x = x.astype(np.uint64)
    c = np.zeros(len(x), dtype=np.int64)
    for b in range(27):
        c += ((x >> np.uint64(b)) & np.uint64(1)).astype(np.int64)
    return c


def main():
    b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = b['coords_v_b'].astype(np.int64)
    stride = 1
    while stride <= 16:
        m = masks(coords, stride)
        rules = popcount(m).sum()
        srt = m[np.argsort(rank_key(m), kind='stable')]
        line = 's%-2d %7d rows %5.2f rules/row |' % (stride, len(m), rules / len(m))
        for R in (16, 32, 64, 128, 256):
            pad = (-len(srt)) % R
            t = np.concatenate([srt, np.zeros(pad, dtype=srt.dtype)]).reshape(-1, R)
            union = np.bitwise_or.reduce(t, axis=1)
            act = popcount(union)
            line += '  R=%-3d %5.2f act, fill %.2f |' % (R, act.mean(), rules / (act.sum() * R))
        print(line, flush=True)
        nxt = coords.copy()
        nxt[:, :3] = (nxt[:, :3] // (stride * 2)) * (stride * 2)
        coords = np.unique(nxt, axis=0)
        stride *= 2


if __name__ == '__main__':
    main()
