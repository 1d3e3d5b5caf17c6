"""Full-size checks (BASELINE.json configs[1]: a ~120 k-point SemanticKITTI-shaped scan, ~83 k
voxels) through size-independent properties, plus the edge cases of the operator set: empty and
single-voxel inputs, ragged batches, the coordinate range limit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def scan():
    from lidal_amd import synth
    b = synth.make_train_batch(n_frames=2, n_points=120000, seed=7122)
    return (torch.from_numpy(b['coords_v_b']).to(DEV), torch.from_numpy(b['feats_v_b']).to(DEV))


def test_fullsize_kernel_map_invariants(scan):
    from lidal_amd.nn import functional as F
    coords, _ = scan
    n = coords.shape[0]
    assert n > 150000
    h = F.sphash(coords)
    assert torch.unique(h).numel() == n                      # distinct voxels -> distinct hashes
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    nbr = kmap.nbr_out
    assert torch.equal(nbr[13], torch.arange(n, device=DEV, dtype=torch.int))   # centre = identity
    assert torch.equal(kmap.nbr_in, nbr.flip(0))             # (i,j,k) <-> (j,i,K-1-k)
    assert int(kmap.nbsizes.sum()) == kmap.total == int((nbr >= 0).sum())
    maps = kmap.nbmaps.long()
    k_of = torch.repeat_interleave(torch.arange(27, device=DEV), kmap.nbsizes.long())
    assert torch.equal(nbr[k_of, maps[:, 1]].long(), maps[:, 0])                # rules match table
    d = coords[maps[:, 0]][:, :3] - coords[maps[:, 1]][:, :3]                   # in = out + offset
    from lidal_amd.nn.utils import get_kernel_offsets
    assert torch.equal(d, get_kernel_offsets(3, 1, 1, device=DEV)[k_of])
    order = kmap.order_out
    assert torch.equal(torch.sort(order.perm.long())[0], torch.arange(n, device=DEV))
    assert torch.equal(order.table, nbr[:, order.perm.long()])


def test_fullsize_strided_transposed_round_trip(scan):
    import lidal_amd
    from lidal_amd.nn import functional as F
    coords, feats = scan
    x = lidal_amd.SparseTensor(feats, coords, 1)
    x.cmaps[(1, 1, 1)] = coords
    g = torch.Generator().manual_seed(0)
    down = F.conv3d(x, torch.randn(8, 4, 32, generator=g).to(DEV), 2, stride=2)
    assert down.s == (2, 2, 2) and (down.C[:, :3] % 2 == 0).all()
    keys = down.C[:, 3].long() * 2 ** 42 + down.C[:, 0].long() * 2 ** 28 + down.C[:, 1].long() * 2 ** 14 + down.C[:, 2].long()
    assert (keys[1:] > keys[:-1]).all()                      # sorted by (b,x,y,z), no duplicates
    km = x.kmaps[((1, 1, 1), (2, 2, 2), (2, 2, 2), (1, 1, 1))]
    assert km.total == coords.shape[0]                       # every fine voxel has one parent
    assert torch.equal((km.nbr_in >= 0).sum(0), torch.ones(coords.shape[0], device=DEV, dtype=torch.long))
    up = F.conv3d(down, torch.randn(8, 32, 32, generator=g).to(DEV), 2, stride=2, transposed=True)
    assert up.s == (1, 1, 1) and up.C is coords and up.F.shape == (coords.shape[0], 32)


def test_fullsize_conv_linearity_and_determinism(scan):
    import lidal_amd
    from lidal_amd.nn import functional as F
    coords, _ = scan
    n = coords.shape[0]
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(27, 32, 64, generator=g) * 0.05).to(DEV)
    a = torch.randn(n, 32, generator=g).to(DEV)
    b = torch.randn(n, 32, generator=g).to(DEV)
    x = lidal_amd.SparseTensor(a, coords, 1)
    ya = F.conv3d(x, w, 3).F
    x.feats = b
    yb = F.conv3d(x, w, 3).F
    x.feats = 2.0 * a - 3.0 * b
    yc = F.conv3d(x, w, 3).F
    err = (yc - (2.0 * ya - 3.0 * yb)).abs().max() / yc.abs().max()
    assert err < 1e-5, err
    x.feats = a
    assert torch.equal(F.conv3d(x, w, 3).F, ya)              # bitwise reproducible
    one = torch.zeros(n, 32, device=DEV)                     # a single occupied input voxel
    one[12345, 5] = 1.0
    x.feats = one
    y1 = F.conv3d(x, w, 3).F
    km = x.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    hit = (km.nbr_out == 12345).nonzero()                    # (k, out_row) pairs fed by that voxel
    exp = torch.zeros_like(y1)
    exp[hit[:, 1]] = w[hit[:, 0], 5]
    assert torch.allclose(y1, exp, atol=1e-6)


def test_fullsize_point_voxel_properties(scan):
    from lidal_amd import PointTensor
    from lidal_amd.network.glue import initial_voxelize, point_to_voxel, voxel_to_point
    coords, feats = scan
    z = PointTensor(feats, coords.float())
    x0 = initial_voxelize(z, 0.05, 0.05)
    assert x0.C.shape == coords.shape                        # the dataset already voxelised: 1:1
    rows = x0.C[:, 3].long() * 2 ** 42 + x0.C[:, 0].long() * 2 ** 28 + x0.C[:, 1].long() * 2 ** 14 + x0.C[:, 2].long()
    ref = coords[:, 3].long() * 2 ** 42 + coords[:, 0].long() * 2 ** 28 + coords[:, 1].long() * 2 ** 14 + coords[:, 2].long()
    assert torch.equal(torch.sort(rows)[0], torch.sort(ref)[0])
    z0 = voxel_to_point(x0, z)
    # stride-1 trilinear ~ identity gather: (C*0.05f)/0.05f is C up to one f32 ulp (4.9e-4 above
    # 4096), so the own-voxel weight is >= 0.999 rather than exactly 1
    w = z.weights[(1, 1, 1)]
    assert torch.allclose(w.sum(1), torch.ones_like(w[:, 0]), atol=1e-5)
    assert w[:, 0].min() > 0.999
    assert (z0.F - feats).abs().max() < 2e-3 * feats.abs().max()
    back = point_to_voxel(x0, z0)
    assert (back.F - x0.F).abs().max() < 2e-3 * x0.F.abs().max()


def test_edge_cases_empty_single_and_range_limit():
    import lidal_amd
    from lidal_amd.nn import functional as F
    empty = torch.zeros((0, 4), dtype=torch.int, device=DEV)
    assert F.sphash(empty).shape == (0,)
    assert F.unique_sorted(torch.zeros(0, dtype=torch.int64, device=DEV)).numel() == 0
    assert F.spcount(torch.zeros(0, dtype=torch.int, device=DEV), 3).tolist() == [0, 0, 0]
    assert F.sphashquery(torch.tensor([5], device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV)).tolist() == [-1]
    one = torch.tensor([[8191, 0, 8191, 3]], dtype=torch.int, device=DEV)      # range limit, one voxel
    kmap, oc = F.build_kernel_map(one, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    assert kmap.total == 1 and kmap.nbmaps.tolist() == [[0, 0]] and kmap.nbsizes.tolist()[13] == 1
    w = torch.randn(27, 4, 32, device=DEV)
    y = F.conv3d(lidal_amd.SparseTensor(torch.ones(1, 4, device=DEV), one, 1), w, 3)
    assert torch.allclose(y.F[0], w[13].sum(0), atol=1e-5)
    d = F.spdownsample(one, 2, 2, 1)
    assert d.tolist() == [[8190, 0, 8190, 3]]
    # ragged batch: frames of very different sizes, batch ids not starting at 0
    g = torch.Generator().manual_seed(2)
    c = torch.cat([torch.cat([torch.randint(0, 40, (n, 3), generator=g), torch.full((n, 1), b)], 1)
                   for n, b in ((3000, 1), (7, 4), (1, 9))]).int()
    c = torch.unique(c, dim=0)
    from oracle.tsref.nn import functional as RF
    nb, ns, sizes, ocr, res = RF.build_kmap(c, (1, 1, 1), (2, 2, 2), (2, 2, 2))
    km, oc = F.build_kernel_map(c.to(DEV), (1, 1, 1), (2, 2, 2), (2, 2, 2))
    assert torch.equal(oc.cpu(), ocr) and torch.equal(km.nbmaps.cpu().long(), nb)


def test_edge_cases_of_the_fused_ops():
    """Empty / single-row / all-ignored inputs through the fused row-wise ops, the dense path and
    the inference epilogues."""
    import lidal_amd
    from lidal_amd.nn import functional as F
    from lidal_amd.nn.functional.dense import rows_linear
    z = torch.zeros((0, 32), device=DEV)
    assert F.add_relu(z, z).shape == (0, 32)
    lab = torch.full((64,), 255, dtype=torch.int64, device=DEV)
    logits = torch.randn(64, 19, device=DEV, requires_grad=True)
    loss = F.cross_entropy(logits, lab, 255)            # nothing to average over: nan, as torch
    assert torch.isnan(loss)
    lab[3] = 7
    loss = F.cross_entropy(logits, lab, 255)
    loss.backward()
    ref = torch.nn.functional.cross_entropy(logits.detach(), lab, ignore_index=255)
    assert abs(loss.item() - ref.item()) < 1e-5 and (logits.grad[lab == 255] == 0).all()
    w = torch.randn(24, 32, device=DEV)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        assert rows_linear(z, w).shape == (0, 24)
        one = rows_linear(torch.ones(1, 32, device=DEV), w)             # one row through the MFMA tile
    assert torch.allclose(one.float()[0], w.sum(1), atol=0.15)
    # inference epilogue on a single voxel at the coordinate range limit
    c1 = torch.tensor([[8191, 0, 8191, 3]], dtype=torch.int, device=DEV)
    wk = torch.randn(27, 8, 32, device=DEV)
    scale, shift = torch.full((32,), 2.0, device=DEV), torch.full((32,), -1.0, device=DEV)
    res = torch.ones(1, 32, device=DEV)
    with torch.no_grad():
        y = F.conv3d(lidal_amd.SparseTensor(torch.ones(1, 8, device=DEV), c1, 1), wk, 3,
                     epilogue=(scale, shift, 2, res)).F
    assert torch.allclose(y[0], torch.relu(wk[13].sum(0) * 2 - 1 + 1), atol=1e-4)


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_fullsize_model_forward_matches_oracle(name):
    """Model-level parity AT the benchmarked size: one ~120 k-point scan (~83 k voxels) through the
    whole network in f32 on the HIP path against the CPU oracle (oracle/models_ref.py, asserted
    bit-identical to the reference's network/*.py by make_golden.py) at 1e-4; then the same
    forward under bf16 autocast (the bench dtype): the arg-max class must agree on > 90 % of the
    voxels and the logits stay within bf16-sized error."""
    import lidal_amd
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet
    from oracle import tsref
    from oracle.models_ref import MinkUNetRef, SPVCNNRef
    from weights import fill_state_dict
    b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
    coords, feats = torch.from_numpy(b['coords_v_b']), torch.from_numpy(b['feats_v_b'])
    assert coords.shape[0] > 70000
    ref_model = fill_state_dict({'spvcnn': SPVCNNRef, 'minkunet': MinkUNetRef}[name](19)).eval()
    torch.set_num_threads(min(32, torch.get_num_threads() * 4))
    with torch.no_grad():
        ref_logits, ref_feat = ref_model(tsref.SparseTensor(feats.clone(), coords.clone()))
    model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).to(DEV).eval()
    with torch.no_grad():
        logits, feat = model(lidal_amd.SparseTensor(feats.to(DEV), coords.to(DEV)))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits16, _ = model(lidal_amd.SparseTensor(feats.to(DEV), coords.to(DEV)))

    def rel(a, ref):
        return ((a.double().cpu() - ref.double()).abs().max() / ref.double().abs().max()).item()
    assert rel(logits, ref_logits) < 1e-4, rel(logits, ref_logits)
    assert rel(feat, ref_feat) < 1e-4
    assert (logits.argmax(1).cpu() == ref_logits.argmax(1)).float().mean() > 0.9999
    agree = (logits16.argmax(1).cpu() == ref_logits.argmax(1)).float().mean().item()
    print("bf16 argmax agreement %.4f, logit rel %.4f" % (agree, rel(logits16.float(), ref_logits)))
    assert agree > 0.9 and rel(logits16.float(), ref_logits) < 0.08, (agree, rel(logits16.float(), ref_logits))
