"""GPU input voxelisation + collate against the fixture produced by the
REFERENCE's own SK_Dataset.__getitem__ / collate_fn (tests/golden/voxelize_small.npz) and against
the CPU oracle at full size.  Integer outputs bit-exact; features to 1 f32 ulp (numpy's matmul may
fuse multiply-adds where the kernel rounds each product)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ulp_close(a, b):
    return np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= np.spacing(np.abs(b).astype(np.float32)).astype(np.float64))


def test_voxelize_matches_reference_dataset(golden_dir):
    from lidal_amd import data
    g = np.load(os.path.join(golden_dir, 'voxelize_small.npz'))
    samples = []
    for i in range(2):
        cv, fv, ui, inv = data.voxelize_scan(torch.from_numpy(g['points%d' % i]).to(DEV),
                                             torch.from_numpy(g['intensity%d' % i]).to(DEV),
                                             g['trans_m%d' % i], g['rnd%d' % i])
        assert cv.dtype == torch.int32 and inv.dtype == torch.int64
        assert np.array_equal(cv.cpu().numpy(), g['coords_v%d' % i])
        assert np.array_equal(inv.cpu().numpy(), g['inverse%d' % i])
        assert _ulp_close(fv.cpu().numpy(), g['feats_v%d' % i])
        # first-occurrence index: coords of the indexed points are the unique rows
        samples.append({'coords_v': cv, 'feats_v': fv, 'inverse_idxs': inv})
    col = data.collate(samples)
    assert np.array_equal(col['coords_v_b'].cpu().numpy(), g['coords_v_b'])
    assert np.array_equal(col['inverse_indices_b'].cpu().numpy(), g['inverse_indices_b'])
    assert _ulp_close(col['feats_v_b'].cpu().numpy(), g['feats_v_b'])


def test_voxelize_full_size_matches_oracle_and_feeds_the_model():
    import lidal_amd
    from lidal_amd import data, synth
    from lidal_amd.nn import functional as F
    from oracle import voxelize_ref
    rng = np.random.default_rng(0)
    pts, inten = synth.raycast_scan(synth.make_world(3), (30.0, 0.0), rng)
    assert pts.shape[0] > 100000
    rs = np.random.RandomState(7)
    trans_m, rnd = data.draw_augmentation(rs)
    cv_r, fv_r, ui_r, inv_r = voxelize_ref.voxelize_scan(pts, inten, trans_m, rnd)
    cv, fv, ui, inv = data.voxelize_scan(torch.from_numpy(pts).to(DEV), torch.from_numpy(inten).to(DEV),
                                         trans_m, rnd)
    assert np.array_equal(cv.cpu().numpy(), cv_r) and np.array_equal(inv.cpu().numpy(), inv_r)
    assert np.array_equal(ui.cpu().numpy(), ui_r)
    assert _ulp_close(fv.cpu().numpy(), fv_r)
    # properties: rows strictly increasing lexicographically, inverse consistent
    k = cv[:, 0].long() * 2 ** 26 + cv[:, 1].long() * 2 ** 13 + cv[:, 2].long()
    assert (k[1:] > k[:-1]).all() and int(inv.max()) == cv.shape[0] - 1
    col = data.collate([{'coords_v': cv, 'feats_v': fv, 'inverse_idxs': inv}])
    kmap, _ = F.build_kernel_map(col['coords_v_b'], (1, 1, 1), (3, 3, 3), (1, 1, 1))
    assert kmap.total > cv.shape[0]


def test_voxelize_rejects_points_outside_the_grid():
    from lidal_amd import data
    pts = torch.tensor([[0.0, 0, 0], [500.0, 0, 0]], device=DEV)      # 500 m * 20 > 8192 voxels
    with pytest.raises(AssertionError, match='not valid'):
        data.voxelize_scan(pts, torch.zeros(2, device=DEV), np.eye(3), np.full(6, 0.5))


def test_register_scan_matches_reference_kdtree_data(golden_dir, tmp_path):
    """World-frame coordinates bit-exact against the data of the KDTree the reference's
    prepare_kdtree_sk.process_frame pickled; pose parsing against the same text files."""
    from lidal_amd import data
    g = np.load(os.path.join(golden_dir, 'register_small.npz'))
    (tmp_path / 'calib.txt').write_text(str(g['calib_txt']))
    (tmp_path / 'poses.txt').write_text(str(g['poses_txt']))
    calib = data.parse_calibration(str(tmp_path / 'calib.txt'))
    poses = data.parse_poses(str(tmp_path / 'poses.txt'), calib)
    assert np.array_equal(poses[0], g['pose'])
    world = data.register_scan(torch.from_numpy(g['points']).to(DEV), poses[0])
    assert world.dtype == torch.float64
    assert np.array_equal(world.cpu().numpy(), g['world'])
    # and it feeds the scorer's frame bank
    from lidal_amd.score import FrameBank
    bank = FrameBank(0.1)
    bank.add(world, torch.full((world.shape[0], 19), 1 / 19, device=DEV))
    assert bank.grid(0).numel() > 0


def test_collate_feeds_the_validation_path():
    """The val collate of the reference returns labels_p_b next to inverse_indices_b
    (sk_dataset.py:227-238) and evaluate.py:95-124 consumes exactly that: voxelize two scans on the
    device, collate them WITH per-point labels, run evaluate_batches, and compare the confusion
    matrix with the reference formula (utils/iou_sk.py:14-19) on the model's own logits."""
    from lidal_amd import SparseTensor, data as ldata, synth
    from lidal_amd.evaluate import evaluate_batches
    from lidal_amd.network import MinkUNet
    rng = np.random.default_rng(12)
    world = synth.make_world(5)
    samples, labels = [], []
    for i in range(2):
        pts, inten = synth.raycast_scan(world, (18.0 + 2 * i, 0.0), rng, n_beams=16, n_az=256)
        trans_m, rnd = ldata.draw_augmentation(np.random.RandomState(40 + i))
        cv, fv, _, inv = ldata.voxelize_scan(torch.from_numpy(pts).to(DEV), torch.from_numpy(inten).to(DEV),
                                             trans_m, rnd)
        lab = torch.from_numpy(rng.integers(0, 19, pts.shape[0])).to(DEV)
        lab[::7] = 255                                           # ignored by the metric (>= 100)
        samples.append({'coords_v': cv, 'feats_v': fv, 'inverse_idxs': inv, 'labels_p': lab})
        labels.append(lab)
    batch = ldata.collate(samples)
    assert batch['labels_p_b'].shape == batch['inverse_indices_b'].shape
    assert torch.equal(batch['labels_p_b'], torch.cat(labels))
    torch.manual_seed(3)
    model = MinkUNet(19).to(DEV)
    conf, ious, miou = evaluate_batches(model, [batch])
    with torch.no_grad():
        logits, _ = model.eval()(SparseTensor(batch['feats_v_b'], batch['coords_v_b']))
    pred = logits[batch['inverse_indices_b']].argmax(1).cpu().numpy()
    gt = batch['labels_p_b'].cpu().numpy()
    keep = gt < 100
    ref = np.bincount(pred[keep] * 19 + gt[keep], minlength=361).reshape(19, 19)
    assert np.array_equal(conf, ref) and conf.sum() == keep.sum()
