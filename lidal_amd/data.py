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

__all__ = ['draw_augmentation', 'voxelize_scan', 'collate', 'parse_calibration', 'parse_poses',
           'register_scan']

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
    feats_p = B.empty((p, 4), torch.float32, dev)
    coords_v = B.empty((max(p, 1), 3), torch.int, dev)
    uniq = B.empty(max(p, 1), torch.int64, dev)
    inverse = B.empty(p, torch.int64, dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)          # [n_out, n_invalid (i32 view)]
    n_invalid = counts[1:].view(torch.int32)[:1]
    ws_bytes = B.lib().lidal_voxelize_points_workspace_bytes(p)
    ws = B.workspace(ws_bytes, dev)
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
    optionally labels_v / labels_p / inverse_idxs.  The batch index becomes the 4th coordinate column and the
    inverse indices are offset by the voxels of the preceding samples."""
    coords, feats, labels, inverse, labels_p = [], [], [], [], []
    off = 0
    for b, s in enumerate(samples):
        c = s['coords_v'].int()
        coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int, device=c.device)], 1))
        feats.append(s['feats_v'].float())
        if 'labels_v' in s:
            labels.append(s['labels_v'].long())
        if 'labels_p' in s:          # val / score modes: per-POINT labels (sk_dataset.py:236-238)
            labels_p.append(s['labels_p'].long())
        if 'inverse_idxs' in s:
            inverse.append(s['inverse_idxs'].long() + off)
            off += c.shape[0]        # == max(inverse) + 1: every voxel is hit by some point
    return {'coords_v_b': torch.cat(coords, 0), 'feats_v_b': torch.cat(feats, 0),
            'labels_v_b': torch.cat(labels, 0) if labels else None,
            'labels_p_b': torch.cat(labels_p, 0) if labels_p else None,
            'inverse_indices_b': torch.cat(inverse, 0) if inverse else None}


# ---- world-frame registration (dataset/prepare_kdtree_sk.py) ---------------------
def _mat_from_values(values):
    pose = np.zeros((4, 4))
    pose[0, 0:4] = values[0:4]
    pose[1, 0:4] = values[4:8]
    pose[2, 0:4] = values[8:12]
    pose[3, 3] = 1.0
    return pose


def parse_calibration(filename):
    """prepare_kdtree_sk.py:39-62: `key: 12 floats` lines -> dict of 4x4 matrices."""
    calib = {}
    with open(filename) as f:
        for line in f:
            key, content = line.strip().split(':')
            calib[key] = _mat_from_values([float(v) for v in content.strip().split()])
    return calib


def parse_poses(filename, calibration):
    """prepare_kdtree_sk.py:10-36: per-scan camera poses -> LiDAR poses Tr^-1 * pose * Tr."""
    tr = calibration['Tr']
    tr_inv = np.linalg.inv(tr)
    poses = []
    with open(filename) as f:
        for line in f:
            pose = _mat_from_values([float(v) for v in line.strip().split()])
            poses.append(np.matmul(tr_inv, np.matmul(pose, tr)))
    return poses


def register_scan(points, pose):
    """prepare_kdtree_sk.py:76-80: sensor-frame points f32 [P,3] (GPU) + 4x4 pose (host f64) ->
    world-frame f64 [P,3], the data the reference hands to sklearn's KDTree (:83); here it goes to
    lidal_amd.score.FrameBank.add, which builds the uniform NN grid."""
    B.require_gpu(points)
    points = points.contiguous().float()
    p = points.shape[0]
    pose_dev = torch.from_numpy(np.ascontiguousarray(pose, dtype=np.float64).reshape(16)).to(points.device)
    world = torch.empty((p, 3), dtype=torch.float64, device=points.device)
    B.check(B.lib().lidal_register_points(B.ptr(points), p, B.ptr(pose_dev), B.ptr(world), B.stream()),
            'register_points')
    return world
