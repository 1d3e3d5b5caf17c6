"""CPU study (no GPU): what does sorting rows by occupancy mask only WITHIN blocks of B rows of
the memory order cost in active offsets per 128-row tile, and what does it buy in gather locality?
Bench batch (5 scans x 120k pts), level 0, 3x3x3 map.  Memory order = the dataset's (per scan
np.unique(axis=0): lexicographic x,y,z)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lidal_amd import synth

def build_nbr(coords):
    c = coords.astype(np.int64)
    key = ((c[:, 3] << 42) | (c[:, 0] << 28) | (c[:, 1] << 14) | c[:, 2])
    order = np.argsort(key, kind='stable'); skey = key[order]
    n = len(c); nbr = np.full((27, n), -1, np.int32)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = ((c[:, 3] << 42) | ((c[:, 0] + dx) << 28) | ((c[:, 1] + dy) << 14) | (c[:, 2] + dz))
                ok = (c[:, 0] + dx >= 0) & (c[:, 1] + dy >= 0) & (c[:, 2] + dz >= 0)
                pos = np.searchsorted(skey, q); pos[pos >= n] = n - 1
                hit = ok & (skey[pos] == q)
                nbr[k, hit] = order[pos[hit]]
                k += 1
    return nbr

def gray_key(nbr):
    rank = np.zeros(27, np.int64); pos = 26
    for want in (3, 2, 1, 0):
        for k in range(27):
            a, b, c = k % 3, (k // 3) % 3, k // 9
            if (a != 1) + (b != 1) + (c != 1) == want:
                rank[k] = pos; pos -= 1
    m = np.zeros(nbr.shape[1], np.int64)
    for k in range(27):
        m |= (nbr[k] >= 0).astype(np.int64) << rank[k]
    g = m.copy()
    for s in (1, 2, 4, 8, 16):
        g ^= g >> s
    return g, m

def evaluate(nbr, perm, label):
    n = nbr.shape[1]; tiles = (n + 127) // 128
    occ = (nbr[:, perm] >= 0)
    pad = tiles * 128 - n
    occ = np.concatenate([occ, np.zeros((27, pad), bool)], 1).reshape(27, tiles, 128)
    act = occ.any(2).sum(0)                                   # active offsets per tile
    # 16-row group activity (what MFMA work is actually issued with wave-level skip, G=1)
    grp = occ.reshape(27, tiles, 8, 16).any(3)
    mfma_groups = grp.sum()
    rules = occ.sum()
    # gather footprint: distinct input rows per tile, and span in memory (rows) of the gathered set
    tab = nbr[:, perm]
    tab = np.concatenate([tab, np.full((27, pad), -1, np.int32)], 1).reshape(27, tiles, 128)
    sample = np.linspace(0, tiles - 1, min(tiles, 400)).astype(int)
    distinct, span = [], []
    for t in sample:
        v = tab[:, t, :].ravel(); v = v[v >= 0]
        u = np.unique(v); distinct.append(len(u)); span.append(u.max() - u.min() + 1 if len(u) else 0)
    # distinct rows per group of 64 consecutive tiles (what an XCD's L2 would see for neighbouring WGs)
    d64 = []
    for t0 in range(0, tiles - 64, max(64, (tiles // 40) // 64 * 64 or 64)):
        v = tab[:, t0:t0 + 64, :].ravel(); v = v[v >= 0]; d64.append(len(np.unique(v)) / len(v))
    print('%-14s act/tile %5.2f  groups16 %.3f of dense (useful %.3f)  distinct rows/tile %6.1f (of %5.1f gathered)  '
          'median span %8d rows  distinct/gathered over 64 tiles %.3f'
          % (label, act.mean(), mfma_groups / (27 * tiles * 8), rules / (mfma_groups * 16),
             np.mean(distinct), rules / tiles, int(np.median(span)), np.mean(d64)))
    return act.mean()

if __name__ == '__main__':
    order = sys.argv[1] if len(sys.argv) > 1 else 'dataset'
    b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = b['coords_v_b']
    if order == 'hash':                                        # SPVCNN level 0: sorted-hash = random order
        rng = np.random.default_rng(0); coords = coords[rng.permutation(len(coords))]
    nbr = build_nbr(coords)
    n = nbr.shape[1]
    print('rows', n, 'rules', int((nbr >= 0).sum()), 'order', order)
    g, m = gray_key(nbr)
    evaluate(nbr, np.arange(n), 'unsorted')
    for B in (512, 1024, 2048, 4096, 8192, 16384, 65536):
        blk = np.arange(n) // B
        perm = np.lexsort((g, blk))
        evaluate(nbr, perm, 'block %d' % B)
    evaluate(nbr, np.argsort(g, kind='stable'), 'global sort')
