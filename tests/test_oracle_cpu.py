"""CPU tests of the oracle itself: spec-derived known answers, algebraic properties that do not
depend on the restatement, gradcheck, and the committed fixtures produced by the REFERENCE's
own code (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import scoring_ref, tsref
from oracle.tsref.nn import functional as RF
from oracle.tsref.nn.utils import get_kernel_offsets


def test_sphash_known_answers():
    c = torch.tensor([[0, 0, 0, 0], [1, 2, 3, 0], [4095, 4096, 8191, 4], [100, 200, 300, 7]],
                     dtype=torch.int)
    assert RF.sphash(c).tolist() == [947293587111810033, 1043245732202901914,
                                     482871551030584986, 15305659009498132]
    # pure-Python FNV-1a-64 of one row, folded to 60 bits
    h = 14695981039346656037
    for w in (100, 200, 300, 7):
        h = ((h ^ w) * 1099511628211) % 2 ** 64
    assert ((h >> 60) ^ (h & 0xFFFFFFFFFFFFFFF)) == 15305659009498132
    off = get_kernel_offsets(3, 1)
    assert torch.equal(RF.sphash(c, off)[13], RF.sphash(c))            # centre offset
    shifted = c.clone()
    shifted[:, :3] += off[5]
    assert torch.equal(RF.sphash(c, off)[5], RF.sphash(shifted))


def test_kernel_offsets_order():
    o3 = get_kernel_offsets(3).tolist()
    assert o3[0] == [-1, -1, -1] and o3[1] == [0, -1, -1] and o3[13] == [0, 0, 0] and o3[26] == [1, 1, 1]
    o2 = get_kernel_offsets(2, 4).tolist()
    assert o2 == [[0, 0, 0], [0, 0, 4], [0, 4, 0], [0, 4, 4], [4, 0, 0], [4, 0, 4], [4, 4, 0], [4, 4, 4]]
    assert all(get_kernel_offsets(3)[26 - k].tolist() == (-get_kernel_offsets(3)[k]).tolist()
               for k in range(27))


def test_hashquery_first_occurrence_and_miss():
    refs = torch.tensor([5, 9, 5, 7], dtype=torch.int64)
    q = torch.tensor([[9, 5], [1, 7]], dtype=torch.int64)
    assert RF.sphashquery(q, refs).tolist() == [[1, 0], [-1, 3]]
    assert RF.sphashquery(q, torch.zeros(0, dtype=torch.int64)).tolist() == [[-1, -1], [-1, -1]]


def test_conv_single_voxel_reproduces_weight_rows():
    c = torch.tensor([[5, 5, 5, 0]], dtype=torch.int)
    w = torch.randn(27, 4, 6)
    out = RF.conv3d(tsref.SparseTensor(torch.eye(4)[:1], c, 1), w, 3)
    assert torch.allclose(out.F, w[13, :1])


def test_conv_dense_grid_equals_torch_conv3d():
    g = torch.Generator().manual_seed(9)
    D, ci, co = 6, 3, 5
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(D), torch.arange(D), indexing='ij')
    c = torch.stack([xx, yy, zz, torch.zeros_like(xx)], -1).reshape(-1, 4).int()
    feats = torch.randn(c.shape[0], ci, generator=g, dtype=torch.float64)
    w = torch.randn(27, ci, co, generator=g, dtype=torch.float64)
    out = RF.conv3d(tsref.SparseTensor(feats, c, 1), w, 3).F
    vol = feats.reshape(D, D, D, ci).permute(3, 0, 1, 2)[None]
    wt = w.reshape(3, 3, 3, ci, co).permute(4, 3, 0, 1, 2)
    dense = torch.nn.functional.conv3d(vol, wt, padding=1)[0].permute(1, 2, 3, 0).reshape(-1, co)
    assert torch.allclose(out, dense, atol=1e-10)


def test_strided_then_transposed_round_trips_coordinates():
    g = torch.Generator().manual_seed(1)
    c = torch.unique(torch.cat([torch.randint(0, 20, (300, 3), generator=g),
                                torch.randint(0, 2, (300, 1), generator=g)], 1).int(), dim=0)
    x = tsref.SparseTensor(torch.randn(c.shape[0], 4, generator=g), c, 1)
    x.cmaps[(1, 1, 1)] = c
    down = RF.conv3d(x, torch.randn(8, 4, 4, generator=g), 2, stride=2)
    assert down.s == (2, 2, 2) and torch.equal(down.C, RF.spdownsample(c, 2, 2, 1))
    assert torch.equal(down.C[:, :3] % 2, torch.zeros_like(down.C[:, :3]))
    up = RF.conv3d(down, torch.randn(8, 4, 4, generator=g), 2, stride=2, transposed=True)
    assert up.s == (1, 1, 1) and torch.equal(up.C, c)
    km = x.kmaps[((1, 1, 1), (2, 2, 2), (2, 2, 2), (1, 1, 1))]
    assert int(km[1].sum()) == c.shape[0]          # every fine voxel has exactly one parent


def test_voxelize_devoxelize_properties_and_gradcheck():
    g = torch.Generator().manual_seed(2)
    idx = torch.tensor([0, 2, 2, -1, 1, 2])
    counts = RF.spcount(idx.int(), 3)
    assert counts.tolist() == [1, 1, 3]
    f = torch.randn(6, 3, generator=g, dtype=torch.float64, requires_grad=True)
    v = RF.spvoxelize(f, idx, counts)
    assert torch.allclose(v[2], f[[1, 2, 5]].mean(0))
    assert torch.autograd.gradcheck(lambda t: RF.spvoxelize(t, idx, counts), (f,))
    coords = torch.rand(5, 4, generator=g) * 8
    iq = torch.randint(0, 4, (8, 5), generator=g)
    w = RF.calc_ti_weights(coords, iq, 2)
    assert torch.allclose(w.sum(0), torch.ones(5), atol=1e-5)
    iq[3, :] = -1
    assert (RF.calc_ti_weights(coords, iq, 2)[3] == 0).all()
    vf = torch.randn(4, 3, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(
        lambda t: RF.spdevoxelize(t, iq.t().contiguous(), w.t().double().contiguous()), (vf,))
    wk = torch.randn(27, 3, 2, generator=g, dtype=torch.float64, requires_grad=True)
    c = torch.unique(torch.randint(0, 4, (20, 4), generator=g).int(), dim=0)
    ff = torch.randn(c.shape[0], 3, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(
        lambda a, b: RF.conv3d(tsref.SparseTensor(a, c, 1), b, 3).F, (ff, wk))


def test_scoring_oracle_reproduces_reference_worker_func(golden_dir):
    g = np.load(os.path.join(golden_dir, 'scoring_small.npz'))
    probs, worlds = list(g['probs']), list(g['worlds'])
    for i in (0, 5, 13, 26):
        d, e, n, c = scoring_ref.score_frame(i, probs, worlds, list(g['sv2point'][i]),
                                             int(g['nei_num']), float(g['dis_thresh']))
        assert np.array_equal(d, g['sv_interds'][i]) and np.array_equal(e, g['sv_interes'][i])
        assert np.array_equal(n, g['sv_pnums'][i]) and np.array_equal(c, g['sv_centers'][i])


def test_neighbour_window_rules():
    assert scoring_ref.neighbour_ids(0, 30, 24) == list(range(13, 25)) + list(range(1, 13))
    ids = scoring_ref.neighbour_ids(29, 30, 24)
    assert ids[:12] == list(range(28, 16, -1)) and ids[12:] == list(range(16, 4, -1))
    for i in range(30):
        ids = scoring_ref.neighbour_ids(i, 30, 24)
        assert len(set(ids)) == 24 and i not in ids and min(ids) >= 0 and max(ids) <= 29


def test_selection_oracle_reproduces_reference_main(golden_dir):
    g = np.load(os.path.join(golden_dir, 'selection_small.npz'))
    out = scoring_ref.select(g['flags_in'], g['sv_interds'], g['sv_interes'], g['sv_pnums'],
                             g['sv_centers'], int(g['train_point_num']))
    assert np.array_equal(out, g['flags_out'])
    assert (out == 1).sum() > 0 and (out == 2).sum() > 0


def test_models_ref_reproduces_reference_model_files(golden_dir):
    """oracle/models_ref.py vs the golden logits of the unchanged reference network/*.py."""
    from oracle.models_ref import MinkUNetRef, SPVCNNRef
    from weights import fill_state_dict
    g = np.load(os.path.join(golden_dir, 'model_small.npz'))
    for name, cls in (('spvcnn', SPVCNNRef), ('minkunet', MinkUNetRef)):
        m = fill_state_dict(cls(19)).eval()
        with torch.no_grad():
            lo, fe = m(tsref.SparseTensor(torch.from_numpy(g['feats']), torch.from_numpy(g['coords'])))
        assert np.array_equal(lo.numpy(), g[name + '_logits'])
        assert np.array_equal(fe.numpy(), g[name + '_feat'])


def test_voxelize_oracle_reproduces_reference_dataset(golden_dir):
    """oracle/voxelize_ref.py vs the fixture of the reference's SK_Dataset.__getitem__."""
    from oracle import voxelize_ref
    g = np.load(os.path.join(golden_dir, 'voxelize_small.npz'))
    for i in range(2):
        cv, fv, ui, inv = voxelize_ref.voxelize_scan(g['points%d' % i], g['intensity%d' % i],
                                                     g['trans_m%d' % i], g['rnd%d' % i])
        assert np.array_equal(cv, g['coords_v%d' % i]) and np.array_equal(inv, g['inverse%d' % i])
        assert np.array_equal(fv, g['feats_v%d' % i])


def _linear_field_case(stride, n_pts=400, seed=0):
    """A dense block of voxels at tensor stride `stride` carrying the affine field f = a x + b y + c z + d
    (distinct a, b, c: a wrong corner order cannot cancel), and points strictly inside the block."""
    g = torch.Generator().manual_seed(seed)
    D = 5
    ax = torch.arange(D) * stride
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing='ij')
    vox = torch.stack([xx, yy, zz, torch.zeros_like(xx)], -1).reshape(-1, 4).int()
    vox = vox[torch.randperm(vox.shape[0], generator=g)]           # row order must not matter
    coef = torch.tensor([0.37, -1.3, 2.9])
    field = (vox[:, :3].float() * coef).sum(1, keepdim=True) + 0.5
    pts = torch.rand(n_pts, 3, generator=g) * (D - 1) * stride * 0.999
    pts = torch.cat([pts, torch.zeros(n_pts, 1)], 1)
    want = (pts[:, :3] * coef).sum(1, keepdim=True) + 0.5
    return vox, field, pts, want


def test_trilinear_devoxelize_reproduces_a_linear_field():
    """Independent pin of the even-kernel corner order: get_kernel_offsets(2, s) enumerates the 8 corners
    z-fastest and calc_ti_weights numbers its weights 4 dx + 2 dy + dz (network/utils.py:69-83); only if
    the two agree does trilinear interpolation reproduce an affine field exactly."""
    for stride in (1, 2, 4, 8):
        vox, field, pts, want = _linear_field_case(stride)
        off = get_kernel_offsets(2, stride, 1)
        floor = torch.cat([torch.floor(pts[:, :3] / stride).int() * stride, pts[:, -1:].int()], 1)
        idx = RF.sphashquery(RF.sphash(floor, off), RF.sphash(vox))                # [8, N]
        assert (idx >= 0).all()
        w = RF.calc_ti_weights(pts, idx, scale=stride).transpose(0, 1).contiguous()
        assert torch.allclose(w.sum(1), torch.ones(pts.shape[0]), atol=1e-6)
        got = RF.spdevoxelize(field, idx.transpose(0, 1).contiguous(), w)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-4 * stride), (stride, (got - want).abs().max())
