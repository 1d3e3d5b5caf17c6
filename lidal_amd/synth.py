"""Synthetic SemanticKITTI-shaped inputs for the hot path (host-side numpy).

There is no dataset and no network on the build or GPU box, so bench.py, smoke() and the
tests drive the path with scans made here:

  * raycast_scan(): a 64-beam x 2048-azimuth spinning LiDAR (elevation -24.8..+2 deg,
    range 2..80 m, 1 cm range noise) ray-cast into a street scene (ground plane, building
    walls on both sides, boxes for cars, rough "vegetation" blobs) -> P ~ 120 k points.
  * voxelize_scan(): the reference input pipeline restated step by step
    (/root/reference/dataset/sk_dataset.py:143-171): random affine `I + 0.1 N(0,1)` with random
    x-flip and z-rotation, feats[:, :3] = transformed metres, feats[:, 3] = intensity,
    x20 (0.05 m voxels), random translation into [0, 8192)^3, astype(int),
    np.unique(axis=0, return_index, return_inverse).
  * collate(): sk_dataset.py:188-242 -- batch index appended as the 4th coordinate column,
    inverse indices offset per sample.
  * make_sequence(): frames along a straight ego path with world-frame coordinates
    (dataset/prepare_kdtree_sk.py:67-88 output contract) and 20 equal-size angular-sector
    supervoxels per frame (stand-in for dataset/prepare_supervoxel_kmeans_sk.py:9-22, whose
    k_means_constrained dependency is absent; only the (sv_id, sv2point) format matters).
"""
import math

import numpy as np

SCALE = 20               # sk_dataset.py:56  (1 / 0.05 m)
FULL_SCALE = 8192        # sk_dataset.py:56
SENSOR_H = 1.73


def make_world(seed=7122, length=400.0):
    """Axis-aligned boxes [B,6] = (xmin,ymin,zmin,xmax,ymax,zmax) in world metres."""
    rng = np.random.default_rng(seed)
    boxes = []
    x = -60.0
    while x < length + 60.0:                       # building fronts on both sides of the road
        for side in (-1.0, 1.0):
            w = rng.uniform(8.0, 25.0)
            d = rng.uniform(6.0, 15.0)
            setback = rng.uniform(9.0, 16.0)
            h = rng.uniform(4.0, 15.0)
            y0 = side * setback
            y1 = side * (setback + d)
            boxes.append([x, min(y0, y1), -SENSOR_H, x + w, max(y0, y1), h])
        x += rng.uniform(10.0, 28.0)
    n_cars = int(length / 6.0)
    for _ in range(n_cars):                        # parked cars / clutter
        cx = rng.uniform(-40.0, length + 40.0)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(3.0, 8.0)
        l, w, h = rng.uniform(3.5, 5.0), rng.uniform(1.6, 2.0), rng.uniform(1.4, 2.0)
        boxes.append([cx - l / 2, cy - w / 2, -SENSOR_H, cx + l / 2, cy + w / 2, -SENSOR_H + h])
    n_veg = int(length / 3.0)
    for _ in range(n_veg):                         # vegetation: many small boxes around a trunk
        cx = rng.uniform(-40.0, length + 40.0)
        cy = rng.choice([-1.0, 1.0]) * rng.uniform(5.0, 9.0)
        for _ in range(12):
            s = rng.uniform(0.2, 0.7)
            px, py, pz = cx + rng.normal(0, 0.9), cy + rng.normal(0, 0.9), rng.uniform(0.0, 4.0)
            boxes.append([px - s, py - s, pz - s, px + s, py + s, pz + s])
    return np.asarray(boxes, dtype=np.float64)


def raycast_scan(world, origin, rng, n_beams=64, n_az=2048, n_points=None):
    """Returns (points f32 [P,3] in the sensor frame, intensity f32 [P])."""
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    az = np.linspace(0.0, 2.0 * math.pi, n_az, endpoint=False) + rng.uniform(0, 2 * math.pi / n_az)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :],
                  np.broadcast_to(se, (n_beams, n_az))], axis=-1).reshape(-1, 3)
    o = np.asarray(origin, dtype=np.float64)
    t_hit = np.full(d.shape[0], np.inf)
    down = d[:, 2] < -1e-9                          # ground plane z = -SENSOR_H (sensor at z=0)
    t_hit[down] = (-SENSOR_H - 0.0) / d[down, 2]
    near = world[(world[:, 3] > o[0] - 85) & (world[:, 0] < o[0] + 85)]
    t_hit = t_hit.reshape(n_beams, n_az)
    d3 = d.reshape(n_beams, n_az, 3)
    inv3 = 1.0 / np.where(np.abs(d3) < 1e-12, 1e-12, d3)
    o3 = np.array([o[0], o[1], 0.0])
    az0, daz = az[0], 2.0 * math.pi / n_az
    for b in near:                                  # slab test, only on the box's azimuth span
        cx = np.array([b[0], b[0], b[3], b[3]]) - o[0]
        cy = np.array([b[1], b[4], b[1], b[4]]) - o[1]
        if cx.min() <= 0.0 <= cx.max() and cy.min() <= 0.0 <= cy.max():
            cols = np.arange(n_az)
        else:
            ang = np.arctan2(cy, cx)
            ref = ang[0]
            rel = np.mod(ang - ref + math.pi, 2.0 * math.pi) - math.pi   # box spans < pi
            a_lo, a_hi = ref + rel.min(), ref + rel.max()
            i_lo = int(math.floor((a_lo - az0) / daz)) - 1
            i_hi = int(math.ceil((a_hi - az0) / daz)) + 1
            cols = np.mod(np.arange(i_lo, i_hi + 1), n_az)
        lo = (b[:3] - o3) * inv3[:, cols]
        hi = (b[3:] - o3) * inv3[:, cols]
        tmin = np.minimum(lo, hi).max(axis=2)
        tmax = np.maximum(lo, hi).min(axis=2)
        ok = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
        cur = t_hit[:, cols]
        t_hit[:, cols] = np.where(ok & (tmin < cur), tmin, cur)
    t_hit = t_hit.reshape(-1)
    t_hit = t_hit + rng.normal(0.0, 0.01, size=t_hit.shape)
    keep = (t_hit > 2.0) & (t_hit < 80.0)
    pts = (d[keep] * t_hit[keep, None]).astype(np.float32)
    inten = rng.uniform(0.0, 1.0, size=pts.shape[0]).astype(np.float32)
    if n_points is not None and pts.shape[0] > n_points:
        sel = np.sort(rng.choice(pts.shape[0], n_points, replace=False))
        pts, inten = pts[sel], inten[sel]
    return pts, inten


def voxelize_scan(points, intensity, rng):
    """sk_dataset.py:98-101,143-171 ('score'/'train' common part).
    Returns coords_v int64 [N,3], feats_v f32 [N,4], unique_idxs, inverse_idxs i64 [P]."""
    raw = np.concatenate([points, intensity[:, None]], axis=1).astype(np.float32)
    feats_p = np.zeros_like(raw)
    coords_p = raw[:, :3]
    feats_p[:, 3] = raw[:, 3]
    trans_m = np.eye(3) + rng.standard_normal((3, 3)) * 0.1
    trans_m[0][0] *= rng.integers(0, 2) * 2 - 1
    theta = rng.random() * 2 * math.pi
    trans_m = np.matmul(trans_m, [[math.cos(theta), math.sin(theta), 0],
                                  [-math.sin(theta), math.cos(theta), 0], [0, 0, 1]])
    coords_p = np.matmul(coords_p, trans_m)
    feats_p[:, :3] = coords_p
    coords_p = coords_p * SCALE
    full = np.array([FULL_SCALE] * 3)
    cmin, cmax = coords_p.min(0), coords_p.max(0)
    offset = (-cmin + np.clip(full - cmax + cmin - 0.001, 0, None) * rng.random(3)
              + np.clip(full - cmax + cmin + 0.001, None, 0) * rng.random(3))
    coords_p = coords_p + offset
    valid = (coords_p.min(1) >= 0) * (coords_p.max(1) < FULL_SCALE)
    assert valid.sum() == len(valid), 'input voxels are not valid'
    coords_v = coords_p.astype(int)
    _, unique_idxs, inverse_idxs = np.unique(coords_v, axis=0, return_index=True,
                                             return_inverse=True)
    return (coords_v[unique_idxs], feats_p[unique_idxs], unique_idxs,
            np.asarray(inverse_idxs).reshape(-1).astype(np.int64))


def collate(samples):
    """sk_dataset.py:188-242.  samples: list of dicts with coords_v, feats_v and optionally
    labels_v / inverse_idxs.  Returns numpy arrays: coords_v_b i32 [N,4] (x,y,z,batch),
    feats_v_b f32 [N,4], labels_v_b i64 [N] | None, inverse_indices_b i64 | None."""
    coords, feats, labels, inverse = [], [], [], []
    inv_off = 0
    for b, s in enumerate(samples):
        c = np.asarray(s['coords_v']).astype(np.int32)
        coords.append(np.concatenate([c, np.full((c.shape[0], 1), b, np.int32)], axis=1))
        feats.append(np.asarray(s['feats_v'], dtype=np.float32))
        if 'labels_v' in s:
            labels.append(np.asarray(s['labels_v'], dtype=np.int64))
        if 'inverse_idxs' in s:
            inv = np.asarray(s['inverse_idxs'], dtype=np.int64)
            inverse.append(inv + inv_off)
            inv_off = int(inverse[-1].max()) + 1
    return {
        'coords_v_b': np.concatenate(coords, 0),
        'feats_v_b': np.concatenate(feats, 0),
        'labels_v_b': np.concatenate(labels, 0) if labels else None,
        'inverse_indices_b': np.concatenate(inverse, 0) if inverse else None,
    }


def make_train_batch(n_frames=1, n_points=120000, seed=7122, n_classes=19, ignore_frac=0.1):
    """A collated training batch: n_frames scans at different ego positions."""
    rng = np.random.default_rng(seed)
    world = make_world(seed)
    samples = []
    for f in range(n_frames):
        pts, inten = raycast_scan(world, (10.0 + 7.0 * f, 0.0), rng, n_points=n_points)
        coords_v, feats_v, uniq, _ = voxelize_scan(pts, inten, rng)
        labels_p = rng.integers(0, n_classes, size=pts.shape[0]).astype(np.int64)
        labels_p[rng.random(pts.shape[0]) < ignore_frac] = 255
        samples.append({'coords_v': coords_v, 'feats_v': feats_v, 'labels_v': labels_p[uniq]})
    return collate(samples)


def make_score_batch(points, intensity, rng, inf_reps=8):
    """dataset/sk_dataloader.py:199-203: `inf_reps` augmented views of ONE frame, collated."""
    samples = []
    for _ in range(inf_reps):
        coords_v, feats_v, _, inverse = voxelize_scan(points, intensity, rng)
        samples.append({'coords_v': coords_v, 'feats_v': feats_v, 'inverse_idxs': inverse})
    return collate(samples)


def angular_supervoxels(points, n_sv=20):
    """20 equal-size sectors by azimuth -> list of int64 index arrays (the sv2point format of
    prepare_supervoxel_kmeans_sk.py:62-74)."""
    order = np.argsort(np.arctan2(points[:, 1], points[:, 0]), kind='stable')
    return [np.sort(chunk).astype(np.int64) for chunk in np.array_split(order, n_sv)]


def make_sequence(n_frames, n_points=120000, seed=7122, step=1.0, n_sv=20, n_beams=64, n_az=2048,
                  start=0, total=None):
    """Frames [start, start + n_frames) of a sequence of `total` frames along a straight ego path
    (every frame has its own RNG stream, so ranks can generate disjoint blocks of one sequence).
    Returns a list of dicts with points f32 [P,3] (sensor frame), intensity f32 [P],
    world f64 [P,3], sv_id i64 [n_sv] (global ids, frame-major), sv2point (list of i64 arrays)."""
    total = total if total is not None else start + n_frames
    world = make_world(seed, length=max(400.0, total * step + 100.0))
    frames = []
    for f in range(start, start + n_frames):
        rng = np.random.default_rng([seed, f])
        ox = 10.0 + step * f
        pts, inten = raycast_scan(world, (ox, 0.0), rng, n_beams=n_beams, n_az=n_az,
                                  n_points=n_points)
        wc = pts.astype(np.float64) + np.array([ox, 0.0, 0.0])
        frames.append({'points': pts, 'intensity': inten, 'world': wc,
                       'sv_id': np.arange(f * n_sv, (f + 1) * n_sv, dtype=np.int64),
                       'sv2point': angular_supervoxels(pts, n_sv)})
    return frames
