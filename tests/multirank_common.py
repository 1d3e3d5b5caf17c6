"""Inputs shared by tests/test_multirank_gpu.py and its child ranks: everything is derived from
fixed seeds so that every process builds bit-identical models, frames and batches."""
import numpy as np
import torch

NEI, REPS, N_FRAMES, N_POINTS = 10, 8, 13, 2500
GRAD_KEYS = ['stem.0.kernel', 'stage2.1.net.0.kernel', 'stage4.2.net.3.kernel', 'up1.0.net.0.kernel',
             'up4.1.1.net.3.kernel', 'classifier.0.weight', 'stage1.0.net.1.weight',
             'point_transforms.1.0.weight']


def make_model(dev, classes=19):
    from lidal_amd.network import SPVCNN
    torch.manual_seed(7122)
    return SPVCNN(classes).to(dev)


def make_frames():
    from lidal_amd import synth
    frames = synth.make_sequence(N_FRAMES, n_points=N_POINTS, seed=41, step=0.05, n_beams=16, n_az=512)
    rng = np.random.default_rng(17)
    out = []
    for i, f in enumerate(frames):
        sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=REPS)
        out.append({'coords': sb['coords_v_b'], 'feats': sb['feats_v_b'], 'inverse': sb['inverse_indices_b'],
                    'world': f['world'], 'sv2point': f['sv2point'],
                    'sv_id': np.arange(i * len(f['sv2point']), (i + 1) * len(f['sv2point']), dtype=np.int64)})
    return out


def to_device(frame, dev):
    from lidal_amd.score import interframe
    ptr, idx, _ = interframe.sv_csr(frame['sv2point'], dev)
    return {'coords': torch.from_numpy(frame['coords']).to(dev), 'feats': torch.from_numpy(frame['feats']).to(dev),
            'inverse': torch.from_numpy(frame['inverse']).to(dev), 'world': torch.from_numpy(frame['world']).to(dev),
            'sv_ptr': ptr, 'sv_idx': idx}


def make_half_batches():
    from lidal_amd import synth
    out = []
    for r in range(2):
        b = synth.make_train_batch(n_frames=1, n_points=3000, seed=900 + r)
        out.append({'coords': torch.from_numpy(b['coords_v_b']), 'feats': torch.from_numpy(b['feats_v_b']),
                    'labels': torch.from_numpy(b['labels_v_b'])})
    return out


def make_val_batches(n=3):
    """Validation batches in the reference's val-collate form (per-point labels + inverse map)."""
    import numpy as np
    from lidal_amd import synth
    out = []
    for i in range(n):
        rng = np.random.default_rng(950 + i)
        pts, inten = synth.raycast_scan(synth.make_world(950 + i), (10.0 + 3.0 * i, 0.0), rng, n_points=3000)
        coords_v, feats_v, _, inverse = synth.voxelize_scan(pts, inten, rng)
        b = synth.collate([{'coords_v': coords_v, 'feats_v': feats_v, 'inverse_idxs': inverse}])
        labels_p = rng.integers(0, 19, size=pts.shape[0]).astype(np.int64)
        labels_p[rng.random(pts.shape[0]) < 0.1] = 255
        out.append({'coords_v_b': torch.from_numpy(b['coords_v_b']), 'feats_v_b': torch.from_numpy(b['feats_v_b']),
                    'inverse_indices_b': torch.from_numpy(b['inverse_indices_b']),
                    'labels_p_b': torch.from_numpy(labels_p)})
    return out
