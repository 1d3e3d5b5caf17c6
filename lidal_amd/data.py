"""Input voxelisation and collation on the GPU: the host-side mirror of the reference's
dataset/sk_dataset.py:143-171 (per-scan augmentation + voxelisation) and :188-242 (collate_fn).
SURVEY.md section 8f, row 1: the step immediately before the hot path.

The random draws stay on the host in the reference's order (`draw_augmentation`), so a seeded
numpy generator reproduces the reference's augmentation; everything per point runs in
lidal_voxelize_points (affine in f64, x20, translation, int cast, unique rows with first-occurrence
index and inverse map).
"""
import math

import numpy as np
import torch

from . import backend as B

__all__ = ['draw_augmentation', 'voxelize_scan', 'collate']

SCALE = 20                # sk_dataset.py:56
FULL_SCALE = 8192


def draw_augmentation(rng=np.random):
    """sk_dataset.py:144-147 and :156 -- the five random draws of one __getitem__ call, in order.
    `rng` is the numpy global module (as the reference uses) or a RandomState."""
    trans_m = np.eye(3) + rng.randn(3, 3) * 0.1
    trans_m[0][0] *= rng.randint(0, 2) * 2 - 1
    theta = rng.rand() * 2 * math.pi
    trans_m = np.matmul(trans_m, [[math.cos(theta), math.sin(theta), 0],
                                  [-math.sin(theta), math.cos(theta), 0], [0, 0, 1]])
    rnd = np.concatenate([rng.rand(3), rng.rand(3)])
    return trans_m, rnd


def voxelize_scan(points, intensity, trans_m, rnd, scale=SCALE, full_scale=FULL_SCALE):
    """points f32 [P,3], intensity f32 [P] on the GPU; trans_m 3x3 and rnd [6] host float64.
    Returns (coords_v i32 [N,3], feats_v f32 [N,4], unique_idxs i64 [N], inverse_idxs i64 [P])
    with numpy's np.unique(axis=0) semantics (rows sorted lexicographically, first occurrence)."""
    B.require_gpu(points, intensity)
    points = points.contiguous().float()
    intensity = intensity.contiguous().float()
    p = points.shape[0]
    dev = points.device
    m_dev = torch.from_numpy(np.ascontiguousarray(trans_m, dtype=np.float64).reshape(9)).to(dev)
    r_dev = torch.from_numpy(np.ascontiguousarray(rnd, dtype=np.float64).reshape(6)).to(dev)
    feats_p = torch.empty((p, 4), dtype=torch.float32, device=dev)
    coords_v = torch.empty((max(p, 1), 3), dtype=torch.int, device=dev)
    uniq = torch.empty(max(p, 1), dtype=torch.int64, device=dev)
    inverse = torch.empty(p, dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)          # [n_out, n_invalid (i32 view)]
    n_invalid = counts[1:].view(torch.int32)[:1]
    ws_bytes = B.lib().lidal_voxelize_points_workspace_bytes(p)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    B.check(B.lib().lidal_voxelize_points(B.ptr(points), B.ptr(intensity), p, B.ptr(m_dev),
                                          B.ptr(r_dev), float(scale), int(full_scale),
                                          B.ptr(feats_p), B.ptr(coords_v), B.ptr(uniq),
                                          B.ptr(inverse), B.ptr(counts), B.ptr(n_invalid), B.ptr(ws),
                                          ws_bytes, B.stream()), 'voxelize_points')
    host = counts.cpu()
    n_out, bad = int(host[0]), int(host[1:].view(torch.int32)[0])
    assert bad == 0, 'input voxels are not valid'            # sk_dataset.py:161
    uniq = uniq[:n_out]
    return coords_v[:n_out], feats_p[uniq], uniq, inverse


def collate(samples):
    """sk_dataset.py:188-242 on device tensors.  samples: dicts with coords_v [N,3], feats_v and
    optionally labels_v / inverse_idxs.  The batch index becomes the 4th coordinate column and the
    inverse indices are offset by the voxels of the preceding samples."""
    coords, feats, labels, inverse = [], [], [], []
    off = 0
    for b, s in enumerate(samples):
        c = s['coords_v'].int()
        coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int, device=c.device)], 1))
        feats.append(s['feats_v'].float())
        if 'labels_v' in s:
            labels.append(s['labels_v'].long())
        if 'inverse_idxs' in s:
            inverse.append(s['inverse_idxs'].long() + off)
            off += c.shape[0]        # == max(inverse) + 1: every voxel is hit by some point
    return {'coords_v_b': torch.cat(coords, 0), 'feats_v_b': torch.cat(feats, 0),
            'labels_v_b': torch.cat(labels, 0) if labels else None,
            'inverse_indices_b': torch.cat(inverse, 0) if inverse else None}
