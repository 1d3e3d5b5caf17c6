"""CPU restatement of the reference's per-scan input voxelisation.  TEST INFRASTRUCTURE.

Follows /root/reference/dataset/sk_dataset.py:98-101,143-171 line by line (numpy, float64 after
the matmul) with the random draws passed in, and sk_dataset.py:188-242 (collate_fn).  PINNED:
tests/golden/make_golden.py runs the reference's own SK_Dataset.__getitem__ / collate_fn on a
synthetic .bin scan under np.random.seed and asserts this restatement reproduces it bit for bit
(tests/golden/voxelize_small.npz)."""
import numpy as np


def voxelize_scan(points, intensity, trans_m, rnd, scale=20, full_scale=8192):
    raw = np.concatenate([points, intensity[:, None]], axis=1).astype(np.float32)
    feats_p = np.zeros_like(raw)
    coords_p = raw[:, :3]
    feats_p[:, 3] = raw[:, 3]
    coords_p = np.matmul(coords_p, trans_m)
    feats_p[:, :3] = coords_p
    coords_p = coords_p * scale
    full = np.array([full_scale] * 3)
    cmin, cmax = coords_p.min(0), coords_p.max(0)
    offset = (-cmin + np.clip(full - cmax + cmin - 0.001, 0, None) * rnd[:3]
              + np.clip(full - cmax + cmin + 0.001, None, 0) * rnd[3:])
    coords_p = coords_p + offset
    valid = (coords_p.min(1) >= 0) * (coords_p.max(1) < full_scale)
    assert valid.sum() == len(valid), 'input voxels are not valid'
    coords_v = coords_p.astype(int)
    _, unique_idxs, inverse_idxs = np.unique(coords_v, axis=0, return_index=True,
                                             return_inverse=True)
    return coords_v[unique_idxs], feats_p[unique_idxs], unique_idxs, np.asarray(inverse_idxs).reshape(-1)
