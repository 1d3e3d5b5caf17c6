"""Parity of every HIP operator with the CPU oracle (oracle.tsref) on the same seeded inputs.
Integer / index results are compared bit-exactly; f32 features within 1e-4 relative
(BASELINE.json north_star); bf16 within bf16 rounding of the f32 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.tsref.nn import functional as RF          # noqa: E402  (checker only)
from oracle.tsref.nn.utils import get_kernel_offsets as ref_offsets  # noqa: E402

DEV = 'cuda'


def _F():
    from lidal_amd.nn import functional as F
    return F


def _coords(n, extent=60, batches=2, seed=0, stride=1):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(0, extent, (n * 2, 3), generator=g, dtype=torch.int32) * stride
    b = torch.randint(0, batches, (n * 2, 1), generator=g, dtype=torch.int32)
    c = torch.unique(torch.cat([c, b], 1), dim=0)
    c = c[torch.randperm(c.shape[0], generator=g)][:n]
    return c.contiguous()


def _surface_coords(n_side=70, batches=2, seed=0):
    """voxels on a wavy 2-D sheet: ~9 of 27 neighbours occupied, like LiDAR surfaces."""
    g = torch.Generator().manual_seed(seed)
    xs, ys = torch.meshgrid(torch.arange(n_side), torch.arange(n_side), indexing='ij')
    out = []
    for b in range(batches):
        z = (8 + 4 * torch.sin(xs / 7.0 + b) + 3 * torch.cos(ys / 5.0)).floor().int()
        c = torch.stack([xs.int() + 100, ys.int() + 50, z + 20, torch.full_like(z, b)], -1)
        out.append(c.reshape(-1, 4))
    c = torch.cat(out)
    return c[torch.randperm(c.shape[0], generator=g)].contiguous().int()


def _relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('n', [0, 1, 63, 1000, 50001])
def test_sphash(n):
    F = _F()
    c = _coords(max(n, 1), extent=8191, seed=n)[:n]
    assert torch.equal(F.sphash(c.to(DEV)).cpu(), RF.sphash(c))
    for ks, st in ((3, 1), (2, 2), (3, 4)):
        off = ref_offsets(ks, st)
        assert torch.equal(F.sphash(c.to(DEV), off.to(DEV)).cpu(), RF.sphash(c, off))


def test_sphash_known_answers():
    F = _F()
    c = torch.tensor([[0, 0, 0, 0], [1, 2, 3, 0], [4095, 4096, 8191, 4], [100, 200, 300, 7]],
                     dtype=torch.int)
    assert F.sphash(c.to(DEV)).cpu().tolist() == [947293587111810033, 1043245732202901914,
                                                  482871551030584986, 15305659009498132]


def test_sphash_rejects_cpu_and_wrong_dtype():
    F = _F()
    with pytest.raises(RuntimeError):
        F.sphash(torch.zeros((4, 4), dtype=torch.int))
    with pytest.raises(AssertionError):
        F.sphash(torch.zeros((4, 4), dtype=torch.int64, device=DEV))


@pytest.mark.parametrize('nq,nr', [(0, 10), (10, 0), (1000, 1000), (100000, 30000)])
def test_sphashquery(nq, nr):
    F = _F()
    g = torch.Generator().manual_seed(nq + nr)
    refs = torch.randint(0, 2 ** 59, (nr,), generator=g, dtype=torch.int64)
    q = torch.randint(0, 2 ** 59, (nq,), generator=g, dtype=torch.int64)
    if nr and nq:
        q[::2] = refs[torch.randint(0, nr, (q[::2].numel(),), generator=g)]
    out = F.sphashquery(q.to(DEV), refs.to(DEV)).cpu()
    assert torch.equal(out, RF.sphashquery(q, refs))


def test_sphashquery_duplicates_first_wins_and_shape():
    F = _F()
    refs = torch.tensor([5, 9, 5, 7, 9, 9, 1], dtype=torch.int64)
    q = torch.tensor([[9, 5], [1, 2], [7, 9]], dtype=torch.int64)
    out = F.sphashquery(q.to(DEV), refs.to(DEV)).cpu()
    assert out.tolist() == [[1, 0], [6, -1], [3, 1]]
    assert torch.equal(out, RF.sphashquery(q, refs))


@pytest.mark.parametrize('n', [1, 5, 4096, 70001])
def test_unique_sorted(n):
    F = _F()
    g = torch.Generator().manual_seed(n)
    k = torch.randint(0, max(2, n // 2), (n,), generator=g, dtype=torch.int64) * 1234567891
    assert torch.equal(F.unique_sorted(k.to(DEV)).cpu(), torch.unique(k))
    h = RF.sphash(_coords(n, seed=n))
    assert torch.equal(F.unique_sorted(h.to(DEV)).cpu(), torch.unique(h))


@pytest.mark.parametrize('ts', [1, 2, 8])
def test_spdownsample(ts):
    F = _F()
    c = _coords(20000, extent=300, batches=3, seed=ts, stride=ts)
    ref = RF.spdownsample(c, 2, 2, ts)
    out = F.spdownsample(c.to(DEV), 2, 2, ts).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize('ks,stride,ts', [(3, 1, 1), (2, 2, 1), (3, 1, 4), (2, 2, 2)])
def test_kernel_map_bit_exact(ks, stride, ts):
    F = _F()
    c = _surface_coords(60, seed=ks + ts)
    c[:, :3] = c[:, :3] // ts * ts
    c = torch.unique(c, dim=0)
    c = c[torch.randperm(c.shape[0], generator=torch.Generator().manual_seed(1))].contiguous()
    st = (stride,) * 3
    nbmaps, nbsizes, sizes, out_coords, results = RF.build_kmap(c, (ts,) * 3, (ks,) * 3, st)
    kmap, oc = F.build_kernel_map(c.to(DEV), (ts,) * 3, (ks,) * 3, st)
    assert kmap.sizes == sizes
    assert torch.equal(oc.cpu(), out_coords)
    assert torch.equal(kmap.nbsizes.cpu().long(), nbsizes)
    assert torch.equal(kmap.nbmaps.cpu().long(), nbmaps)
    assert torch.equal(kmap.nbr_out.cpu().long(), results)
    assert kmap.total == int(nbsizes.sum())
    # inverse table: nbr_in[k][i] = j  <=>  nbr_out[k][j] = i
    nbr_in = kmap.nbr_in.cpu()
    exp = torch.full_like(nbr_in, -1)
    kk, jj = torch.nonzero(results != -1, as_tuple=True)
    exp[kk, results[kk, jj]] = jj.int()
    assert torch.equal(nbr_in, exp)
    if kmap.symmetric:
        assert torch.equal(nbr_in, kmap.nbr_out.cpu().flip(0))


def test_spcount_voxelize_devoxelize():
    F = _F()
    g = torch.Generator().manual_seed(3)
    n, m = 30000, 7000
    idx = torch.randint(-1, m, (n,), generator=g, dtype=torch.int64)
    counts_ref = RF.spcount(idx.int(), m)
    counts = F.spcount(idx.int().to(DEV), m)
    assert torch.equal(counts.cpu(), counts_ref)
    for c in (4, 32, 96):
        feats = torch.randn(n, c, generator=g)
        f_ref = feats.clone().requires_grad_(True)
        o_ref = RF.spvoxelize(f_ref, idx, counts_ref)
        f_gpu = feats.to(DEV).requires_grad_(True)
        o_gpu = F.spvoxelize(f_gpu, idx.to(DEV), counts)
        assert _relerr(o_gpu.cpu(), o_ref) < 1e-5
        go = torch.randn(m, c, generator=g)
        o_ref.backward(go)
        o_gpu.backward(go.to(DEV))
        assert _relerr(f_gpu.grad.cpu(), f_ref.grad) < 1e-5
    # devoxelize
    idx8 = torch.randint(-1, m, (n, 8), generator=g, dtype=torch.int32)
    w = torch.rand(n, 8, generator=g)
    for c in (32, 96, 256):
        vf = torch.randn(m, c, generator=g)
        v_ref = vf.clone().requires_grad_(True)
        o_ref = RF.spdevoxelize(v_ref, idx8, w)
        v_gpu = vf.to(DEV).requires_grad_(True)
        o_gpu = F.spdevoxelize(v_gpu, idx8.to(DEV), w.to(DEV))
        assert _relerr(o_gpu.cpu(), o_ref) < 1e-5
        go = torch.randn(n, c, generator=g)
        o_ref.backward(go)
        o_gpu.backward(go.to(DEV))
        assert _relerr(v_gpu.grad.cpu(), v_ref.grad) < 1e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('stride,c', [(16, 256), (16, 128), (8, 64), (4, 32)])
def test_devoxelize_backward_through_the_cells(stride, c, dtype):
    """lidal_devoxelize_bwd_cells (the coarse levels of the training step: every gradient row read once) on the corner
    tables of a synthetic scan (network/glue.py corner_tables = utils.py:67-79): against the f64 definition of the
    gradient -- the bars of the per-voxel form -- and against lidal_devoxelize_bwd_sorted; twice for run-to-run
    bit-equality; the structure it relies on (every point of a cell has the same eight corners) checked on the tables."""
    from lidal_amd import PointTensor, SparseTensor, synth
    from lidal_amd import backend as B
    from lidal_amd.network.glue import corner_tables, initial_tables
    from lidal_amd.nn.functional import devoxelize as DV
    from lidal_amd.nn.functional import spdownsample
    from lidal_amd.nn.functional.invlist import inverse_lists, segment_workspace
    L = B.lib()
    dev = torch.device(DEV)
    batch = synth.make_train_batch(n_frames=2, n_points=60000, seed=991)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    z = PointTensor(None, coords.float())
    x0 = SparseTensor(None, initial_tables(z, 0.05, 0.05), 1)
    x0.cmaps.setdefault(x0.stride, x0.coords)
    cs = x0.C
    s = 1
    while s < stride:
        cs = spdownsample(cs, 2, 2, s)
        s *= 2
        x0.cmaps[(s, s, s)] = cs
    xs = SparseTensor(None, cs, stride)
    xs.cmaps, xs.kmaps = x0.cmaps, x0.kmaps
    #  the marker the cell form is gated on is set only for a caller who vouches that every point's own
    # voxel exists in x -- x derived from z, as here and in SPVCNN; a generic voxel_to_point keeps the per-voxel lists
    idx_plain, _ = corner_tables(xs, z)
    assert not getattr(idx_plain, '_lidal_cell_corners', False) and not DV.cells_mode(idx_plain, idx_plain.shape[0], cs.shape[0], c)
    idx8, w8 = corner_tables(xs, z, own_cells=True)
    n, m = idx8.shape[0], cs.shape[0]
    assert idx8 is idx_plain and getattr(idx8, '_lidal_cell_corners', False)
    # the points' own voxel index is column 0 of the corner index (the list F.spvoxelize keeps is shared with the cells)
    from lidal_amd.network.glue import point_tables
    pidx, _ = point_tables(xs, z)
    assert torch.equal(pidx.int(), idx8[:, 0]) and getattr(idx8, '_lidal_cell_index', None) is not None
    # the structure: the corners of a point are the corners of its cell's first point
    vorder, vseg, corder, cseg = DV.devox_cells(idx8, m)
    first = vorder[vseg[:-1]].long()
    assert bool((idx8[:, 0] >= 0).all()) and torch.equal(idx8, idx8[first][idx8[:, 0].long()])
    assert int(vseg[-1]) == n
    g = torch.Generator(device='cpu').manual_seed(stride * 1000 + c)
    gout = torch.randn(n, c, generator=g).to(dev).to(dtype)
    code = B.dtype_code(dtype)
    nb = L.lidal_devoxelize_bwd_cells_workspace_bytes(m, c)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        gin = torch.full((m, c), float('nan'), device=dev).to(dtype)
        B.check(L.lidal_devoxelize_bwd_cells(B.ptr(gout), B.ptr(vorder), B.ptr(vseg), B.ptr(w8), B.ptr(corder), B.ptr(cseg),
                                             B.ptr(gin), m, c, code, B.ptr(ws), nb, B.stream()), 'cells')
        outs.append(gin)
    assert torch.equal(outs[0], outs[1])
    # f64 definition
    ref = torch.zeros(m, c, dtype=torch.float64, device=dev)
    gd = gout.double()
    for j in range(8):
        ok = idx8[:, j] >= 0
        ref.index_add_(0, idx8[ok, j].long(), gd[ok] * w8[ok, j].double().unsqueeze(1))
    scale = float(ref.abs().max())
    err = float((outs[0].double() - ref).abs().max())
    assert err < (1e-2 if dtype == torch.bfloat16 else 2e-5) * scale, (err, scale)
    # the per-voxel form on the same operands
    order, seg = inverse_lists(idx8, m, w8)
    wsb, nbs = segment_workspace(8 * n, m, c, dev)
    gin2 = torch.empty((m, c), device=dev, dtype=dtype)
    B.check(L.lidal_devoxelize_bwd_sorted(B.ptr(gout), B.ptr(order), B.ptr(seg), B.ptr(w8), B.ptr(gin2), m, c, code, 8 * n,
                                          B.ptr(wsb), nbs, B.stream()), 'sorted')
    d = (outs[0].double() - gin2.double()).abs()
    # (two associations of f32 sums over up to hundreds of terms: a few 1e-6 of the largest element apart)
    step = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -12) * ref.abs().clamp_min(1e-3 * scale)
    assert float((d / step).max()) <= 1.0 + 1e-6


@pytest.mark.parametrize('scale', [1, 2, 8])
def test_calc_ti_weights(scale):
    F = _F()
    g = torch.Generator().manual_seed(scale)
    n = 20000
    coords = torch.rand(n, 4, generator=g) * 200
    idx = torch.randint(-1, 500, (8, n), generator=g, dtype=torch.int64)
    ref = RF.calc_ti_weights(coords, idx, scale)
    out = F.calc_ti_weights(coords.to(DEV), idx.to(DEV), scale).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-6
    w, i32 = F.ti_weights_and_index(coords.to(DEV), idx.to(DEV), scale)
    assert torch.equal(i32.cpu().long(), idx.t())
    assert (w.cpu() - ref.t()).abs().max().item() < 2e-6


def _conv_case(ci, co, ks, stride, transposed, dtype, seed=0, n_side=48):
    """Run one conv fwd+bwd on GPU and on the oracle (f32).  Returns relative errors."""
    import lidal_amd
    from lidal_amd.nn import functional as F
    from oracle import tsref
    g = torch.Generator().manual_seed(seed)
    c = _surface_coords(n_side, seed=seed)
    if transposed:      # input lives on the coarse level; the map comes from the strided conv
        fine_ref = tsref.SparseTensor(torch.zeros(c.shape[0], 4), c, 1)
        fine_gpu = lidal_amd.SparseTensor(torch.zeros(c.shape[0], 4, device=DEV), c.to(DEV), 1)
        fine_ref.cmaps[(1, 1, 1)] = fine_ref.C      # as the stem conv's output would have
        fine_gpu.cmaps[(1, 1, 1)] = fine_gpu.C
        wd = torch.randn(ks ** 3, 4, 4, generator=g)
        x_ref = RF.conv3d(fine_ref, wd, ks, stride=stride)
        x_gpu = F.conv3d(fine_gpu, wd.to(DEV), ks, stride=stride)
        assert torch.equal(x_gpu.C.cpu(), x_ref.C)
        n = x_ref.C.shape[0]
    else:
        n = c.shape[0]
    feats = torch.randn(n, ci, generator=g)
    fan = ci * ks ** 3
    weight = (torch.rand(ks ** 3, ci, co, generator=g) * 2 - 1) / fan ** 0.5 * 3 ** 0.5
    f_ref = feats.clone().requires_grad_(True)
    w_ref = weight.clone().requires_grad_(True)
    f_gpu = feats.to(DEV).to(dtype).requires_grad_(True)
    w_gpu = weight.to(DEV).to(dtype).requires_grad_(True)
    if transposed:
        x_ref.feats, x_gpu.feats = f_ref, f_gpu
        o_ref = RF.conv3d(x_ref, w_ref, ks, stride=stride, transposed=True)
        o_gpu = F.conv3d(x_gpu, w_gpu, ks, stride=stride, transposed=True)
    else:
        o_ref = RF.conv3d(tsref.SparseTensor(f_ref, c, 1), w_ref, ks, stride=stride)
        o_gpu = F.conv3d(lidal_amd.SparseTensor(f_gpu, c.to(DEV), 1), w_gpu, ks, stride=stride)
    assert torch.equal(o_gpu.C.cpu(), o_ref.C) and o_gpu.s == o_ref.s
    go = torch.randn(o_ref.F.shape, generator=g)
    o_ref.F.backward(go)
    o_gpu.F.backward(go.to(DEV).to(dtype))
    return (_relerr(o_gpu.F.detach().float().cpu(), o_ref.F.detach()),
            _relerr(f_gpu.grad.float().cpu(), f_ref.grad),
            _relerr(w_gpu.grad.float().cpu(), w_ref.grad))


CONV_SHAPES = [(4, 32), (32, 32), (32, 64), (64, 64), (96, 96), (128, 96), (64, 128), (192, 128),
               (256, 256), (384, 256)]


@pytest.mark.parametrize('ci,co', CONV_SHAPES)
def test_conv3d_k3_f32(ci, co):
    errs = _conv_case(ci, co, 3, 1, False, torch.float32, seed=ci + co,
                      n_side=48 if ci * co < 256 * 256 else 30)
    assert max(errs) < 1e-4, errs


@pytest.mark.parametrize('ci,co', [(32, 32), (64, 64), (128, 128), (256, 256)])
def test_conv3d_strided_and_transposed_f32(ci, co):
    errs = _conv_case(ci, co, 2, 2, False, torch.float32, seed=ci)
    assert max(errs) < 1e-4, errs
    errs = _conv_case(ci, co if co != 256 else 128, 2, 2, True, torch.float32, seed=ci + 1)
    assert max(errs) < 1e-4, errs


@pytest.mark.parametrize('ci,co', [(4, 32), (32, 32), (96, 96), (128, 96), (192, 128), (384, 256)])
def test_conv3d_k3_bf16(ci, co):
    errs = _conv_case(ci, co, 3, 1, False, torch.bfloat16, seed=ci + co, n_side=40)
    # inputs, weights, outputs rounded to bf16 (2^-8 relative each), f32 accumulation
    assert max(errs) < 3e-2, errs


def test_conv3d_deterministic_and_1x1():
    import lidal_amd
    from lidal_amd.nn import functional as F
    g = torch.Generator().manual_seed(5)
    c = _surface_coords(40, seed=5).to(DEV)
    feats = torch.randn(c.shape[0], 64, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(27, 64, 96, generator=g) * 0.05).to(DEV).requires_grad_(True)
    outs = []
    for _ in range(2):
        feats.grad = w.grad = None
        o = F.conv3d(lidal_amd.SparseTensor(feats, c, 1), w, 3)
        o.F.square().sum().backward()
        outs.append((o.F.detach().clone(), feats.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b), 'conv fwd/bwd must be bitwise reproducible'
    w1 = torch.randn(64, 32, generator=g).to(DEV)
    o1 = F.conv3d(lidal_amd.SparseTensor(feats.detach(), c, 1), w1, 1)
    assert _relerr(o1.F.cpu(), feats.detach().cpu() @ w1.cpu()) < 1e-4


def test_conv3d_dense_grid_equals_torch_conv3d():
    """Oracle-independent property: on a fully occupied grid a k3 sparse conv equals
    torch.nn.functional.conv3d (interior voxels) with the weight re-laid out x-fastest."""
    import lidal_amd
    from lidal_amd.nn import functional as F
    g = torch.Generator().manual_seed(9)
    D, ci, co = 10, 32, 32
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(D), torch.arange(D), indexing='ij')
    c = torch.stack([xx, yy, zz, torch.zeros_like(xx)], -1).reshape(-1, 4).int()
    feats = torch.randn(c.shape[0], ci, generator=g)
    w = torch.randn(27, ci, co, generator=g) * 0.1
    out = F.conv3d(lidal_amd.SparseTensor(feats.to(DEV), c.to(DEV), 1), w.to(DEV), 3).F.cpu()
    vol = feats.reshape(D, D, D, ci).permute(3, 0, 1, 2)[None]            # [1,ci,z,y,x]
    wt = w.reshape(3, 3, 3, ci, co).permute(4, 3, 0, 1, 2)                # [co,ci,kz,ky,kx]
    dense = torch.nn.functional.conv3d(vol, wt, padding=1)[0].permute(1, 2, 3, 0).reshape(-1, co)
    assert _relerr(out, dense) < 1e-4


def test_voxelize_devoxelize_are_reproducible_and_skewed_lists():
    """The scatter sums run as ordered per-voxel gathers: bitwise identical across runs, also with
    very uneven voxel populations (one voxel receiving thousands of points)."""
    F = _F()
    g = torch.Generator().manual_seed(11)
    n, m, c = 50000, 300, 128
    idx = torch.randint(0, m, (n,), generator=g, dtype=torch.int64)
    idx[:20000] = 7                                   # heavy voxel
    idx[20000:20010] = -1
    counts_ref = RF.spcount(idx.int(), m)
    feats = torch.randn(n, c, generator=g)
    ref = RF.spvoxelize(feats, idx, counts_ref)
    outs = []
    for _ in range(2):
        i_dev = idx.to(DEV)
        outs.append(F.spvoxelize(feats.to(DEV), i_dev, F.spcount(i_dev.int(), m)))
    assert torch.equal(outs[0], outs[1])
    assert _relerr(outs[0].cpu(), ref) < 1e-5
    idx8 = torch.randint(-1, m, (n, 8), generator=g, dtype=torch.int32)
    idx8[:, 0] = 3
    w = torch.rand(n, 8, generator=g)
    w[::3, 2] = 0.0
    vf = torch.randn(m, c, generator=g)
    v_ref = vf.clone().requires_grad_(True)
    go = torch.randn(n, c, generator=g)
    RF.spdevoxelize(v_ref, idx8, w).backward(go)
    grads = []
    for _ in range(2):
        v = vf.to(DEV).requires_grad_(True)
        F.spdevoxelize(v, idx8.to(DEV), w.to(DEV)).backward(go.to(DEV))
        grads.append(v.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert _relerr(grads[0].cpu(), v_ref.grad) < 1e-4


@pytest.mark.parametrize('dtype,c,n', [(torch.float32, 32, 5000), (torch.float32, 96, 70001),
                                       (torch.float32, 256, 1500), (torch.bfloat16, 96, 30000),
                                       (torch.bfloat16, 384, 2049), (torch.float32, 4, 100)])
def test_batch_norm_rows_matches_torch(dtype, c, n):
    """spnn.BatchNorm on the HIP kernels vs torch.nn.BatchNorm1d in f64 on the CPU: output,
    running statistics, dx / dgamma / dbeta; train and eval; large mean / small variance column."""
    import lidal_amd.nn as spnn
    g = torch.Generator().manual_seed(c + n)
    x = torch.randn(n, c, generator=g) * 2 + 0.5
    x[:, 1] = x[:, 1] * 0.05 + 10.0                      # |mean| >> std: shifted sums must hold
    go = torch.randn(n, c, generator=g)
    ref = torch.nn.BatchNorm1d(c).double()
    mine = spnn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
        ref.bias.copy_(torch.randn(c, generator=g) * 0.1)
        mine.weight.copy_(ref.weight.float())
        mine.bias.copy_(ref.bias.float())
    xq = x.to(dtype)                                     # both sides see the same rounded input
    xr = xq.double().requires_grad_(True)
    xg = xq.to(DEV).requires_grad_(True)
    yr = ref(xr)
    yg = mine(xg)
    assert yg.dtype == dtype
    yr.backward(go.to(dtype).double())
    yg.backward(go.to(dtype).to(DEV))
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert _relerr(yg.float().cpu(), yr) < tol
    assert _relerr(xg.grad.float().cpu(), xr.grad) < (2e-4 if dtype == torch.float32 else 3e-2)
    assert _relerr(mine.weight.grad.cpu(), ref.weight.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert _relerr(mine.bias.grad.cpu(), ref.bias.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert _relerr(mine.running_mean.cpu(), ref.running_mean) < 1e-5
    assert _relerr(mine.running_var.cpu(), ref.running_var) < 1e-4
    assert int(mine.num_batches_tracked) == 1
    ref.eval(), mine.eval()
    assert _relerr(mine(xq.to(DEV)).float().cpu(), ref(xq.double())) < tol
    # fused ReLU (the form the model uses): y = relu(bn(x)), backward masks dy where y <= 0
    ref.train(), mine.train()
    mine.fused_relu = True
    xr2 = xq.double().requires_grad_(True)
    xg2 = xq.to(DEV).requires_grad_(True)
    ref.zero_grad(), mine.zero_grad()
    yr2 = torch.relu(ref(xr2))
    yg2 = mine(xg2)
    yr2.backward(go.to(dtype).double())
    yg2.backward(go.to(dtype).to(DEV))
    assert _relerr(yg2.float().cpu(), yr2.detach()) < tol
    assert _relerr(xg2.grad.float().cpu(), xr2.grad) < (2e-4 if dtype == torch.float32 else 3e-2)
    assert _relerr(mine.weight.grad.cpu(), ref.weight.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert _relerr(mine.bias.grad.cpu(), ref.bias.grad) < (2e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batch_norm_backward_reductions_full_size(dtype):
    """dgamma / dbeta / saved statistics of ONE BatchNorm at the bench size (4e5 rows x 96 channels)
    against an f64 reference, PER CHANNEL at 1e-4 -- including channels whose sum(dy * xhat) is
    ill-conditioned (dy nearly uncorrelated with x: the sum is ~1e-3 of the sum of magnitudes) and
    a channel with |mean| >> std.  The kernels accumulate every reduction in f64 (bn.hip), so the
    error left is the f32 rounding of each term: |err| <= 2e-7 * sum|terms| as well."""
    from lidal_amd.nn.functional.norm import batch_norm_rows
    n, c = 400003, 96
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, generator=g) * 1.5 + 0.3
    x[:, 1] = x[:, 1] * 0.02 + 25.0
    go = torch.randn(n, c, generator=g)                      # uncorrelated with x: cancelling sums
    go[:, 2] += 0.5 * x[:, 2]                                # one well-conditioned channel
    xq, gq = x.to(dtype), go.to(dtype)
    w = (torch.rand(c, generator=g) + 0.5)
    b = torch.randn(c, generator=g) * 0.1
    xd, gd = xq.double(), gq.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    xhat = (xd - mean) * torch.rsqrt(var + 1e-5)
    ref_gamma, ref_beta = (gd * xhat).sum(0), gd.sum(0)
    mag_gamma, mag_beta = (gd * xhat).abs().sum(0), gd.abs().sum(0)
    assert (ref_gamma.abs() / mag_gamma).median() < 5e-3     # the fixture IS ill-conditioned
    xg = xq.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    y = batch_norm_rows(xg, wg, bg, rm, rv, True, 0.1, 1e-5)
    y.backward(gq.to(DEV))
    got_gamma, got_beta = wg.grad.double().cpu(), bg.grad.double().cpu()
    # f32 statistics feed xhat: an error dm of the saved mean moves dgamma by dm * invstd * sum(dy),
    # which is part of the budget (dm <= 1 ulp of the mean)
    slack = 1.2e-7 * mean.abs() * torch.rsqrt(var + 1e-5) * ref_beta.abs()
    err_g, err_b = (got_gamma - ref_gamma).abs(), (got_beta - ref_beta).abs()
    assert (err_g <= 2e-7 * mag_gamma + 2 * slack + 1e-4 * ref_gamma.abs()).all(), \
        (err_g / mag_gamma).max()
    assert (err_b <= 2e-7 * mag_beta).all(), (err_b / mag_beta).max()
    ok = ref_gamma.abs() > 1e-3 * mag_gamma                  # 1e-4 relative wherever it is meaningful
    assert ((err_g / ref_gamma.abs())[ok] < 1e-4).all() and ok.sum() >= 1
    assert (err_b / ref_beta.abs().clamp_min(1e-3 * mag_beta) < 1e-4).all()
    assert _relerr(rm.cpu(), 0.1 * mean) < 1e-6
    assert _relerr(rv.cpu(), 0.9 + 0.1 * xd.var(0, unbiased=True)) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dense_rows_matmul_and_linear(dtype):
    """1x1x1 conv / point-branch Linear: forward, dx and the split-K weight gradient vs f64."""
    import lidal_amd.nn as spnn
    from lidal_amd.nn.functional.dense import rows_matmul
    g = torch.Generator().manual_seed(3)
    n, ci, co = 50001, 128, 96
    x = torch.randn(n, ci, generator=g).to(dtype)
    w = (torch.randn(ci, co, generator=g) * 0.1)
    go = torch.randn(n, co, generator=g).to(dtype)
    xr = x.double().requires_grad_(True)
    wr = w.to(dtype).double().requires_grad_(True)
    (xr @ wr).backward(go.double())
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(dtype).to(DEV).requires_grad_(True)
    y = rows_matmul(xg, wg)
    y.backward(go.to(DEV))
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert _relerr(y.float().cpu(), (xr @ wr).detach()) < tol
    assert _relerr(xg.grad.float().cpu(), xr.grad) < tol
    assert _relerr(wg.grad.float().cpu(), wr.grad) < tol
    lin = spnn.Linear(ci, 19).to(DEV)            # 19 classes: padded internally to the vector width
    ref = torch.nn.Linear(ci, 19).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in lin.state_dict().items()})
    xl = x.float().to(DEV).requires_grad_(True)
    out = lin(xl)
    out.square().sum().backward()
    xl_r = x.double().requires_grad_(True)
    ref(xl_r).square().sum().backward()
    assert _relerr(out.detach().cpu(), ref(xl_r).detach()) < 1e-4
    assert _relerr(lin.weight.grad.cpu(), ref.weight.grad) < 1e-4
    assert _relerr(lin.bias.grad.cpu(), ref.bias.grad) < 1e-4
    assert _relerr(xl.grad.cpu(), xl_r.grad) < 1e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_add_relu_and_cross_entropy_match_torch(dtype):
    from lidal_amd.nn.functional.fused import add_relu, cross_entropy
    g = torch.Generator().manual_seed(6)
    a = torch.randn(30001, 96, generator=g).to(dtype)
    b = torch.randn(30001, 96, generator=g).to(dtype)
    go = torch.randn(30001, 96, generator=g).to(dtype)
    ar, br = a.float().clone().requires_grad_(True), b.float().clone().requires_grad_(True)
    yr = torch.relu(ar + br)
    yr.backward(go.float())
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yg = add_relu(ag, bg)
    yg.backward(go.to(DEV))
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert _relerr(yg.float().cpu(), yr.detach()) < tol
    assert torch.equal(ag.grad, bg.grad)
    if dtype == torch.float32:
        assert torch.equal(ag.grad.cpu(), ar.grad)
    else:       # the bf16 sum can round a tiny positive to exactly representable values only
        assert ((ag.grad.float().cpu() != 0) == (ar.grad != 0)).float().mean() > 0.999
    # cross-entropy, ignore_index 255, mean over the labelled rows
    n, c = 50000, 19
    logits = (torch.randn(n, c, generator=g) * 3).to(dtype)
    labels = torch.randint(0, c, (n,), generator=g)
    labels[torch.rand(n, generator=g) < 0.1] = 255
    lr = logits.double().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=255, reduction='mean')
    (ref * 1.7).backward()
    lg = logits.to(DEV).requires_grad_(True)
    loss = cross_entropy(lg, labels.to(DEV), 255)
    (loss * 1.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + (0 if dtype == torch.float32 else 1e-6)
    assert _relerr(lg.grad.float().cpu(), lr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert (lg.grad[labels.to(DEV) == 255] == 0).all()


def test_voxel_exchange_bf16_rows_equal_f32_compute_rounded():
    """bf16 feature rows go through the point<->voxel kernels as bf16 with f32 accumulation: the
    results must equal the f32 kernels run on the widened rows, rounded once at the end."""
    F = _F()
    g = torch.Generator().manual_seed(8)
    n, m, c = 40000, 9000, 32
    idx = torch.randint(0, m, (n,), generator=g)
    counts = F.spcount(idx.int().to(DEV), m)
    x = torch.randn(n, c, generator=g).bfloat16().to(DEV)
    gv = torch.randn(m, c, generator=g).bfloat16().to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = F.spvoxelize(xa, idx.to(DEV), counts)
    ya.backward(gv)
    xb = x.float().requires_grad_(True)
    yb = F.spvoxelize(xb, idx.to(DEV), counts)
    yb.backward(gv.float())
    assert ya.dtype == torch.bfloat16
    assert ((ya.float() - yb).abs() <= 2.0 ** -8 * yb.abs() + 1e-6).all()         # (same note as below)
    assert (ya == yb.bfloat16()).float().mean() > 0.99
    assert torch.equal(xa.grad, xb.grad.bfloat16())
    # devoxelize
    idx8 = torch.randint(-1, m, (n, 8), generator=g).int().to(DEV)
    w8 = torch.rand(n, 8, generator=g).to(DEV)
    w8[idx8 < 0] = 0
    f = torch.randn(m, c, generator=g).bfloat16().to(DEV)
    gp = torch.randn(n, c, generator=g).bfloat16().to(DEV)
    fa = f.clone().requires_grad_(True)
    pa = F.spdevoxelize(fa, idx8, w8)
    pa.backward(gp)
    fb = f.float().requires_grad_(True)
    pb = F.spdevoxelize(fb, idx8, w8)
    pb.backward(gp.float())
    assert pa.dtype == torch.bfloat16 and torch.equal(pa, pb.bfloat16())
    # the ordered sums take 16 bytes of a row per lane: 8 bf16 or 4 f32, so the two element types split a list over
    # different lane groups and add in different orders -- f32 accumulation either way: the bf16 result is the f32
    # one rounded once, up to the last bit where the f32 sums differ by their order
    ga, gb = fa.grad.float(), fb.grad
    assert (ga - gb).abs().max() <= 2.0 ** -8 * gb.abs().max()
    assert ((ga - gb).abs() <= 2.0 ** -8 * gb.abs() + 1e-6).all()
    assert (fa.grad == gb.bfloat16()).float().mean() > 0.99


@pytest.mark.parametrize('m,c', [(50, 256), (700, 96), (3000, 32)])
def test_voxel_exchange_long_contributor_lists(m, c):
    """Coarse levels: hundreds of points per voxel.  The ordered sums are then split over several
    waves / workgroups per voxel; results must match an f64 scatter-add and be reproducible."""
    F = _F()
    g = torch.Generator().manual_seed(m)
    n = 60000
    idx = torch.randint(0, m, (n,), generator=g)
    counts = F.spcount(idx.int().to(DEV), m)
    x = torch.randn(n, c, generator=g)
    xg = x.to(DEV).requires_grad_(True)
    y = F.spvoxelize(xg, idx.to(DEV), counts)
    ref = torch.zeros(m, c, dtype=torch.float64).index_add_(0, idx, x.double())
    ref /= torch.bincount(idx, minlength=m).clamp(min=1).double()[:, None]
    assert _relerr(y.detach().cpu(), ref) < 1e-5
    assert torch.equal(y, F.spvoxelize(xg, idx.to(DEV), counts))
    idx8 = torch.randint(-1, m, (n, 8), generator=g).int()
    w8 = torch.rand(n, 8, generator=g)
    w8[idx8 < 0] = 0
    f = torch.randn(m, c, generator=g).to(DEV).requires_grad_(True)
    gp = torch.randn(n, c, generator=g)
    p = F.spdevoxelize(f, idx8.to(DEV), w8.to(DEV))
    p.backward(gp.to(DEV))
    ref_g = torch.zeros(m, c, dtype=torch.float64)
    for k in range(8):
        ok = idx8[:, k] >= 0
        ref_g.index_add_(0, idx8[ok, k].long(), gp[ok].double() * w8[ok, k, None].double())
    assert _relerr(f.grad.cpu(), ref_g) < 1e-5
    g1 = f.grad.clone()
    f.grad = None
    F.spdevoxelize(f, idx8.to(DEV), w8.to(DEV)).backward(gp.to(DEV))
    assert torch.equal(g1, f.grad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_affine_relu_epilogue(dtype):
    """lidal_conv_apply's inference epilogue == conv followed by the affine map and ReLU."""
    F = _F()
    from lidal_amd import SparseTensor
    from lidal_amd.nn.functional.conv import conv3d
    coords = _surface_coords(60, 2, seed=3).to(DEV)
    n, ci, co = coords.shape[0], 32, 96
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, generator=g).to(dtype).to(DEV)
    w = (torch.randn(27, ci, co, generator=g) * 0.1).to(DEV)
    scale = (torch.rand(co, generator=g) + 0.5).to(DEV)
    shift = torch.randn(co, generator=g).to(DEV)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        plain = conv3d(SparseTensor(x, coords), w, 3).F.float()
        for relu in (False, True):
            fused = conv3d(SparseTensor(x, coords), w, 3, epilogue=(scale, shift, relu)).F.float()
            ref = plain * scale + shift
            if relu:
                ref = torch.relu(ref)
            # `plain` was rounded to the storage dtype before the affine map, `fused` was not
            tol = 1e-6 if dtype == torch.float32 else 2e-2
            assert _relerr(fused.cpu(), ref.cpu()) < tol
            if relu:
                assert (fused >= 0).all()
        # residual operand: relu & 1 acts before the sum, relu & 2 after it (residual block)
        res = torch.randn(n, co, generator=g).to(dtype).to(DEV)
        a = conv3d(SparseTensor(x, coords), w, 3, epilogue=(scale, shift, 1, res)).F.float()
        b = conv3d(SparseTensor(x, coords), w, 3, epilogue=(scale, shift, 2, res)).F.float()
        base = plain * scale + shift
        tol = 1e-6 if dtype == torch.float32 else 2e-2
        assert _relerr(a.cpu(), (torch.relu(base) + res.float()).cpu()) < tol
        assert _relerr(b.cpu(), torch.relu(base + res.float()).cpu()) < tol


def test_batchnorm_module_counts_batches_in_kernel():
    """spnn.BatchNorm1d.num_batches_tracked is incremented by the statistics kernel (no separate
    launch) in training mode only, exactly as nn.BatchNorm1d counts."""
    from lidal_amd import nn as spnn
    bn = spnn.BatchNorm1d(32).to(DEV)
    x = torch.randn(1000, 32, device=DEV)
    bn.train()
    for _ in range(3):
        bn(x)
    assert int(bn.num_batches_tracked.item()) == 3
    bn.eval()
    bn(x)
    with torch.no_grad():
        bn(x)
    assert int(bn.num_batches_tracked.item()) == 3


def test_kernel_map_rules_built_lazily_match_eager():
    """Under no_grad only the neighbour table is built; the torchsparse-order rule lists derived
    from it later (mode 2) must equal the ones built in the same call with gradients enabled."""
    F = _F()
    for ks, st in ((3, 1), (2, 2)):
        c = _surface_coords(50, 2, seed=ks).to(DEV)
        eager, oc_e = F.build_kernel_map(c, (1, 1, 1), (ks,) * 3, (st,) * 3)
        assert eager._rules is not None
        with torch.no_grad():
            lazy, oc_l = F.build_kernel_map(c, (1, 1, 1), (ks,) * 3, (st,) * 3)
        assert lazy._rules is None
        assert torch.equal(oc_e, oc_l) and torch.equal(eager.nbr_out, lazy.nbr_out)
        assert torch.equal(eager.nbsizes, lazy.nbsizes) and torch.equal(eager.koff, lazy.koff)
        assert torch.equal(eager.nbmaps, lazy.nbmaps)


def test_dense_epilogue_affine_relu_residual():
    """Inference epilogue of the dense path: act((x @ W^T + b) * scale + shift) + residual in one
    kernel (bf16 under autocast), and the unfused f32 fallback computing the same expression."""
    from lidal_amd.nn.functional.dense import rows_linear
    g = torch.Generator().manual_seed(21)
    n, ci, co = 5003, 64, 96
    x = torch.randn(n, ci, generator=g).to(DEV)
    w = (torch.randn(co, ci, generator=g) * 0.1).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    scale = (torch.rand(co, generator=g) + 0.5).to(DEV)
    shift = torch.randn(co, generator=g).to(DEV)
    res = torch.randn(n, co, generator=g).to(DEV)
    ref = torch.relu((x.double() @ w.double().t() + b.double()) * scale.double() + shift.double()) + res.double()
    with torch.no_grad():
        y32 = rows_linear(x, w, b, epilogue=(scale, shift, True, res))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y16 = rows_linear(x, w, b, epilogue=(scale, shift, True, res.bfloat16()))
    assert y32.dtype == torch.float32 and _relerr(y32.cpu(), ref.cpu()) < 1e-5
    assert y16.dtype == torch.bfloat16 and _relerr(y16.float().cpu(), ref.cpu()) < 2e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_apply_image_is_bitwise_conv_apply(dtype):
    """The two generations of the conv kernel through their C-ABIs: lidal_conv_apply (weights [k][co][ci]
    staged through registers; rounds 1-3 of the library, since round 4 the test-only object
    tests/native/liblidal_gen1.so) and the shipped lidal_conv_apply_image (weights as LDS images moved by
    LDS-DMA, conv_img.hip) run the same offset and reduction order, so every output bit agrees --
    forward and flipped (data-gradient) walk, epilogue + residual, channel counts off the tile sizes
    (ragged last column block, partial reduction slice), the dense identity form, n_out not a multiple
    of the tile, and a one-row input."""
    import ctypes
    import os
    from lidal_amd import backend as B
    F = _F()
    L = B.lib()
    gen1_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'liblidal_gen1.so')
    assert os.path.exists(gen1_path), 'build it with `python tests/native/build.py` (__graft_entry__.build() does)'
    G1 = ctypes.CDLL(gen1_path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    G1.lidal_conv_apply.restype = i32
    G1.lidal_conv_apply.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    code = B.dtype_code(dtype)
    g = torch.Generator().manual_seed(17)

    def both(x, w_kio, order, k, kflip, ep=None):
        n_in, ci = x.shape
        co = w_kio.shape[2]
        n_out = order.n_rows if order is not None else n_in
        wt = w_kio.permute(0, 2, 1).contiguous().to(dtype)
        nb = L.lidal_conv_weight_image_bytes(k, ci, co, code, n_out)
        img = torch.empty(nb, dtype=torch.uint8, device=DEV)
        B.check(L.lidal_conv_weight_image(B.ptr(w_kio), B.dtype_code(w_kio.dtype), 0, B.ptr(img), code, k, ci,
                                          co, n_out, B.stream()), 'image')
        outs = []
        for fn, wop in ((G1.lidal_conv_apply, wt), (L.lidal_conv_apply_image, img)):
            out = torch.full((n_out, co), float('nan'), dtype=dtype, device=DEV)
            sc, sh, relu, res = ep if ep else (None, None, 0, None)
            tab, prm, tmk = (order.table, order.perm, order.tile_masks) if order is not None else (None,) * 3
            extra = (None,) if fn is L.lidal_conv_apply_image else ()          # tile_stats
            B.check(fn(B.ptr(x), B.ptr(wop), B.ptr(tab), B.ptr(prm), B.ptr(tmk), B.ptr(out), n_in, n_out, ci,
                       co, k, kflip, code, B.ptr(sc), B.ptr(sh), relu, B.ptr(res), *extra, B.stream()), 'conv')
            outs.append(out)
        assert not torch.isnan(outs[1].float()).any()
        assert torch.equal(outs[0], outs[1])
        return outs[1]

    coords = _surface_coords(61, 2, seed=5).to(DEV)                  # 7442 rows: ragged last tile
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    n = coords.shape[0]
    vec = 4 if dtype == torch.float32 else 8
    for ci, co in ((32, 32), (96, 96), (8 * vec // 4, 64), (128, 96), (256, 128), (192, 20), (40, 200)):
        x = torch.randn(n, ci, generator=g).to(dtype).to(DEV)
        w = (torch.randn(27, ci, co, generator=g) * 0.1).to(DEV)
        for kflip in (0, 1):
            both(x, w, kmap.order_out, 27, kflip)
    x = torch.randn(n, 64, generator=g).to(dtype).to(DEV)
    w = (torch.randn(27, 64, 96, generator=g) * 0.1).to(DEV)
    ep = ((torch.rand(96, generator=g) + 0.5).to(DEV), torch.randn(96, generator=g).to(DEV), 3,
          torch.randn(n, 96, generator=g).to(dtype).to(DEV))
    both(x, w, kmap.order_out, 27, 0, ep)
    km2, _ = F.build_kernel_map(coords, (1, 1, 1), (2, 2, 2), (2, 2, 2))      # strided + its transpose
    both(x, (torch.randn(8, 64, 32, generator=g) * 0.1).to(DEV), km2.order_out, 8, 0)
    xc = torch.randn(km2.sizes[1], 32, generator=g).to(dtype).to(DEV)
    both(xc, (torch.randn(8, 32, 64, generator=g) * 0.1).to(DEV), km2.order_in, 8, 0)
    both(x, (torch.randn(1, 64, 20, generator=g) * 0.1).to(DEV), None, 1, 0)                  # dense form
    both(x[:1].contiguous(), (torch.randn(1, 64, 96, generator=g) * 0.1).to(DEV), None, 1, 0)  # one row


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_epilogue_batch_statistics(dtype):
    """conv3d(..., want_stats=True): the kernel leaves (count, mean, M2) per 128-row tile and column
    of the values it stored; merged, they are the batch statistics a pass over the output gives, and
    spnn.BatchNorm fed with them produces what it produces from its own statistics pass (output,
    running statistics, gradients).  Sparse k3 layer with a ragged last tile, a column count off the
    tile width, and the dense (Linear) form."""
    import lidal_amd
    import lidal_amd.nn as spnn
    from lidal_amd.nn.functional.conv import conv3d
    from lidal_amd.nn.functional.dense import rows_linear
    g = torch.Generator().manual_seed(23)
    coords = _surface_coords(61, 2, seed=7).to(DEV)
    n = coords.shape[0]
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        for ci, co in ((32, 96), (64, 40)):
            x = (torch.randn(n, ci, generator=g) + 0.3).to(DEV)
            w = (torch.randn(27, ci, co, generator=g) * 0.1).to(DEV).requires_grad_(True)
            out = conv3d(lidal_amd.SparseTensor(x, coords), w, 3, want_stats=True).F
            if dtype == torch.float32:             # parity mode: BatchNorm keeps its own f64 statistics pass
                assert not hasattr(out, '_lidal_bn_stats')
                continue
            st = out._lidal_bn_stats.double().cpu()
            assert st.shape == (-(-n // 128), co, 3)
            st = st.reshape(co, -(-n // 128), 3).permute(1, 0, 2)       # the buffer is laid out [co][tiles][3] (csrc/conv_img.hip)
            o = out.detach().double().cpu()
            cnt = st[:, :, 0].sum(0)
            mean = (st[:, :, 0] * st[:, :, 1]).sum(0) / cnt
            m2 = (st[:, :, 2] + st[:, :, 0] * (st[:, :, 1] - mean) ** 2).sum(0)
            assert torch.equal(cnt, torch.full((co,), float(n), dtype=torch.float64))
            assert _relerr(mean, o.mean(0)) < 1e-5 and _relerr(m2 / n, o.var(0, unbiased=False)) < 1e-5
            bn_a, bn_b = spnn.BatchNorm1d(co).to(DEV), spnn.BatchNorm1d(co).to(DEV)
            bn_a.fused_relu = bn_b.fused_relu = True
            xa = out.detach().clone().requires_grad_(True)
            xa._lidal_bn_stats = out._lidal_bn_stats
            xb = out.detach().clone().requires_grad_(True)
            from lidal_amd import backend as B
            B.HITS.clear()
            ya, yb = bn_a(xa), bn_b(xb)
            go = torch.randn(n, co, generator=g).to(DEV).to(ya.dtype)
            ya.backward(go), yb.backward(go)
            tol = 1e-5 if dtype == torch.float32 else 1e-2
            assert _relerr(ya.float().cpu(), yb.detach().float().cpu()) < tol
            assert _relerr(bn_a.running_var.cpu(), bn_b.running_var.cpu()) < 1e-5
            assert _relerr(bn_a.running_mean.cpu(), bn_b.running_mean.cpu()) < 1e-5
            assert _relerr(xa.grad.float().cpu(), xb.grad.float().cpu()) < tol
            assert _relerr(bn_a.weight.grad.cpu(), bn_b.weight.grad.cpu()) < 1e-3
        x = torch.randn(5003, 64, generator=g).to(DEV).requires_grad_(True)
        lin = torch.nn.Linear(64, 96).to(DEV)
        y = rows_linear(x, lin.weight, lin.bias, want_stats=True)
        if dtype == torch.bfloat16:
            st = y._lidal_bn_stats.double().cpu()
            st = st.reshape(96, -1, 3).permute(1, 0, 2)                 # [co][tiles][3]
            mean = (st[:, :, 0] * st[:, :, 1]).sum(0) / st[:, :, 0].sum(0)
            assert _relerr(mean, y.detach().double().cpu().mean(0)) < 1e-5


def _wgrad_abi(a, b, pairs, koff, a_col, k):
    """lidal_conv_wgrad through the C-ABI with the library's own scratch plan."""
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.conv import wgrad_scratch
    ca, cb = a.shape[1], b.shape[1]
    gw = torch.full((k, ca, cb), float('nan'), dtype=torch.float32, device=a.device)
    part = wgrad_scratch(a.shape[0], b.shape[0], k, ca, cb, a.dtype, a.device)
    B.check(B.lib().lidal_conv_wgrad(B.ptr(a), B.ptr(b), a.shape[0], b.shape[0], B.ptr(pairs), B.ptr(koff),
                                     a_col, B.ptr(gw), B.ptr(part), part.shape[0], k, ca, cb,
                                     B.dtype_code(a.dtype), B.stream()), 'conv_wgrad')
    return gw


@pytest.mark.parametrize('ca,cb', [(32, 32), (96, 96), (128, 96), (48, 40), (256, 256), (64, 384)])
def test_wgrad_dma_exact_products_of_bf16_operands(ca, cb):
    """The bf16 weight gradient (LDS-DMA gathers, one even run of 64-rule stages per workgroup,
    slabs added in workgroup order; csrc/wgrad_dma.hip) against the f64 sum of the exact products of
    the same bf16 operands, rule list of backend.convolution_backward_cuda: 3x3x3 map (offsets with
    few and with no rules, runs crossing offset boundaries), 2x2x2 strided map in both gather
    directions (a_col 0 / 1), the identity list of the dense layers, and lists shorter than a stage.
    f32 accumulation in a different order than the reference: 2e-5 of the largest entry; two launches
    bitwise equal."""
    F = _F()
    g = torch.Generator().manual_seed(ca * 1000 + cb)
    cases = []
    c = _surface_coords(34, 2, seed=ca + cb).to(DEV)                      # 2312 voxels
    km, _ = F.build_kernel_map(c, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    cases.append(('k3', km._nbmaps_cap, km.koff, 27, c.shape[0], c.shape[0], 0))
    km2, oc = F.build_kernel_map(c, (1, 1, 1), (2, 2, 2), (2, 2, 2))
    cases.append(('k2s2', km2._nbmaps_cap, km2.koff, 8, c.shape[0], oc.shape[0], 0))
    cases.append(('k2s2 transposed', km2._nbmaps_cap, km2.koff, 8, oc.shape[0], c.shape[0], 1))
    tiny = _surface_coords(3, 1, seed=1).to(DEV)                            # 9 voxels: every offset < 64 rules
    km3, _ = F.build_kernel_map(tiny, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    cases.append(('tiny', km3._nbmaps_cap, km3.koff, 27, 9, 9, 0))
    n_dense = 1000
    cases.append(('dense', None, torch.tensor([0, n_dense], dtype=torch.int64, device=DEV), 1, n_dense, n_dense, 0))
    for name, pairs, koff, k, n_a, n_b, a_col in cases:
        a = torch.randn(n_a, ca, generator=g).to(DEV).bfloat16()
        b = torch.randn(n_b, cb, generator=g).to(DEV).bfloat16()
        got = _wgrad_abi(a, b, pairs, koff, a_col, k)
        again = _wgrad_abi(a, b, pairs, koff, a_col, k)
        assert torch.equal(got, again), name
        ko = koff.cpu().tolist()
        ref = torch.zeros(k, ca, cb, dtype=torch.float64, device=DEV)
        for kk in range(k):
            if ko[kk + 1] == ko[kk]:
                continue
            if pairs is None:
                ia = ib = torch.arange(ko[kk], ko[kk + 1], device=DEV)
            else:
                pr = pairs[ko[kk]:ko[kk + 1]].long()
                ia, ib = (pr[:, 1], pr[:, 0]) if a_col else (pr[:, 0], pr[:, 1])
            ref[kk] = a[ia].double().t() @ b[ib].double()
        assert torch.isfinite(got).all(), name
        err = (got.double() - ref).abs().max().item() / ref.abs().max().clamp_min(1e-30).item()
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batch_norm_shortcut_sum_and_relu_ride_in_the_normalising_pass(dtype):
    """The end of a residual block, relu(bn(x) + shortcut) (network/utils.py:171), from the BatchNorm's
    normalising pass: bitwise the output and gradients of BatchNorm -> add_relu."""
    from lidal_amd.nn.functional.fused import add_relu
    from lidal_amd.nn.functional.norm import batch_norm_rows
    g = torch.Generator().manual_seed(12)
    n, c = 4999, 96
    x = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    res = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    go = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    w = (torch.rand(c, generator=g) + 0.5).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    outs = []
    for fused in (False, True):
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        if fused:
            y = batch_norm_rows(xi, wi, bi, rm, rv, True, 0.1, 1e-5, False, None, None, ri, True)
        else:
            y = add_relu(batch_norm_rows(xi, wi, bi, rm, rv, True, 0.1, 1e-5, False), ri)
        y.backward(go)
        outs.append((y.detach(), xi.grad, ri.grad, wi.grad, bi.grad, rm, rv))
    for a, bb in zip(*outs):
        assert torch.equal(a, bb)
    assert (outs[1][0] == 0).any() and (outs[1][0] > 0).any()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batch_norm_residual_rides_in_the_normalising_pass(dtype):
    """y = relu(bn(x)) + residual from ONE pass (the point-branch sum of network/spvcnn.py:104,111,118)
    is bitwise what the stand-alone sum of the normalised rows and the residual gives, statistics
    from a pass over x or from a convolution's tile triples alike; the residual's gradient is
    grad_out itself and the other gradients do not change."""
    from lidal_amd.nn.functional.norm import batch_norm_rows
    g = torch.Generator().manual_seed(11)
    n, c = 5001, 64
    x = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    res = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    go = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    w = (torch.rand(c, generator=g) + 0.5).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    outs = []
    for fused in (False, True):
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        if fused:
            y = batch_norm_rows(xi, wi, bi, rm, rv, True, 0.1, 1e-5, True, None, None, ri)
        else:
            y = batch_norm_rows(xi, wi, bi, rm, rv, True, 0.1, 1e-5, True) + ri
        y.backward(go)
        outs.append((y.detach(), xi.grad, ri.grad, wi.grad, bi.grad, rm, rv))
    for a, bb in zip(*outs):
        assert torch.equal(a, bb)
    assert torch.equal(outs[1][2], go)


@pytest.mark.parametrize('n,bits', [(1, 8), (63, 3), (255, 8), (256, 8), (257, 9), (1000, 27), (4097, 16), (50001, 27),
                                    (396662, 27), (396662, 8), (1000003, 19), (3200000, 17), (70000, 32)])
def test_radix_sort_pairs_is_the_stable_sort(n, bits):
    """csrc/sort.hip (what lidal_kmap_order sorts the 8-bit masks with): keys and values
    bit-equal to torch.sort(stable=True) -- a stable sort has one answer -- for lists shorter than
    a round, of exactly one range, of several ranges, with heavy duplicates (3- and 8-bit keys) and
    with full 32-bit keys; the inputs are left untouched."""
    from lidal_amd import backend as B
    g = torch.Generator().manual_seed(n + bits)
    hi = 1 << bits
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)
    if n > 1000:        # a run of equal keys that spans ranges, and the extreme values
        keys[n // 3: n // 3 + n // 5] = keys[0]
        keys[-1], keys[1] = hi - 1, 0
    vals = torch.randperm(n, generator=g).int()
    k_dev = (keys & 0xFFFFFFFF).to(torch.int64).to(DEV)
    k32 = torch.where(k_dev >= 2 ** 31, k_dev - 2 ** 32, k_dev).int()      # same bits as u32
    v_dev = vals.to(DEV)
    k_in, v_in = k32.clone(), v_dev.clone()
    ko, vo = torch.empty_like(k32), torch.empty_like(v_dev)
    nbytes = B.lib().lidal_sort_pairs_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    B.check(B.lib().lidal_sort_pairs(B.ptr(k32), B.ptr(v_dev), B.ptr(ko), B.ptr(vo), n, bits, B.ptr(ws), nbytes,
                                     B.stream()), 'sort_pairs')
    sk, order = torch.sort(keys.to(DEV), stable=True)
    got_keys = ko.long() & 0xFFFFFFFF
    assert torch.equal(got_keys, sk)
    assert torch.equal(vo, v_dev[order])
    assert torch.equal(k32, k_in) and torch.equal(v_dev, v_in)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_batch_norm_backward_reads_a_channel_slice_in_place(dtype):
    """The up stages concatenate [deconv output | skip]; the BatchNorm behind the deconv receives the
    left channel slice of the concatenation's gradient -- a strided view.  lidal_bn_bwd reads it in
    place (dy_stride); results are bitwise those of a contiguous copy."""
    from lidal_amd.nn.functional.norm import batch_norm_rows
    g = torch.Generator().manual_seed(3)
    n, c, extra = 4099, 96, 32
    x = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    wide = torch.randn(n, c + extra, generator=g).to(DEV).to(dtype)
    w = (torch.rand(c, generator=g) + 0.5).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    outs = []
    for go in (wide[:, :c], wide[:, :c].contiguous()):
        xi = x.clone().requires_grad_(True)
        wi, bi = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = batch_norm_rows(xi, wi, bi, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), True, 0.1, 1e-5, True)
        y.backward(go)
        outs.append((xi.grad, wi.grad, bi.grad))
    assert not wide[:, :c].is_contiguous()
    for a, bb in zip(*outs):
        assert torch.equal(a, bb)


@pytest.mark.parametrize('stride', [1, 2, 4, 8, 16])
def test_floor_coords_is_the_reference_expression(stride):
    """network/utils.py:44-47: `torch.floor(z.C[:, :3] / s).int() * s` + the batch column, in one
    kernel (lidal_floor_coords); bit-equal to the torch expression incl. negative and whole values."""
    from lidal_amd import PointTensor
    from lidal_amd.network.glue import _floor_to_stride
    g = torch.Generator().manual_seed(stride)
    c = (torch.rand(20001, 4, generator=g) * 4000 - 300)
    c[:100, :3] = torch.arange(-50, 50).float()[:, None]            # whole numbers, negatives
    c[:, 3] = torch.randint(0, 5, (20001,), generator=g).float()
    c = c.to(DEV)
    want = torch.cat([torch.floor(c[:, :3] / stride).int() * stride, c[:, -1].int().view(-1, 1)], 1)
    got = _floor_to_stride(PointTensor(torch.zeros(c.shape[0], 1, device=DEV), c), stride)
    assert got.dtype == torch.int32 and torch.equal(got, want)


@pytest.mark.parametrize('stride', [1, 2, 4, 8])
def test_trilinear_devoxelize_reproduces_a_linear_field(stride):
    """The HIP hash / query / ti-weights / devoxelize chain of voxel_to_point (network/utils.py:66-102)
    interpolates an affine field exactly: pins the z-fastest corner order of the 2x2x2 offsets against
    the 4 dx + 2 dy + dz numbering of the weights, independently of the oracle."""
    from test_oracle_cpu import _linear_field_case
    F = _F()
    from lidal_amd.nn.utils import get_kernel_offsets
    vox, field, pts, want = _linear_field_case(stride, n_pts=3000, seed=stride)
    vox, field, pts = vox.to(DEV), field.to(DEV), pts.to(DEV)
    off = get_kernel_offsets(2, stride, 1, device=DEV)
    floor = torch.cat([torch.floor(pts[:, :3] / stride).int() * stride, pts[:, -1:].int()], 1)
    idx = F.sphashquery(F.sphash(floor, off), F.sphash(vox))
    assert (idx >= 0).all()
    w = F.calc_ti_weights(pts, idx, scale=stride).transpose(0, 1).contiguous()
    got = F.spdevoxelize(field, idx.transpose(0, 1).contiguous().int(), w)
    assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=1e-4 * stride), (got.cpu() - want).abs().max()


@pytest.mark.parametrize('n,bits,with_vals', [(1, 64, True), (8191, 60, False), (8192, 63, True), (8193, 39, True),
                                              (396662, 60, False), (396662, 63, True), (120000, 63, True),
                                              (1500000, 48, False)])
def test_radix_sort_u64_is_the_stable_sort(n, bits, with_vals):
    """The 64-bit form of csrc/sort.hip (sorted unique of coordinate hashes / packed coordinates, the
    voxeliser's row keys, the scorer's cell keys): bit-equal to torch.sort(stable=True), keys only or
    with a payload; tiles that end exactly at, one short of and one past a tile boundary."""
    from lidal_amd import backend as B
    g = torch.Generator().manual_seed(n + bits)
    keys = torch.randint(0, 2 ** 62, (n,), generator=g, dtype=torch.int64)
    if bits < 62:
        keys = keys >> (62 - bits)
    elif bits > 62:
        keys = keys | (torch.randint(0, 2, (n,), generator=g, dtype=torch.int64) << 62)
    if n > 1000:
        keys[n // 3: n // 3 + n // 5] = keys[0]           # a long run of equal keys spanning tiles
        keys[-1], keys[1] = (1 << min(bits, 63)) - 1, 0
    vals = torch.randperm(n, generator=g).int().to(DEV)
    k_dev = keys.to(DEV)
    k_in = k_dev.clone()
    ko, vo = torch.empty_like(k_dev), torch.empty_like(vals)
    nbytes = B.lib().lidal_sort_pairs_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    B.check(B.lib().lidal_sort_pairs_u64(B.ptr(k_dev), B.ptr(vals) if with_vals else None, B.ptr(ko),
                                         B.ptr(vo) if with_vals else None, n, bits, B.ptr(ws), nbytes, B.stream()),
            'sort_pairs_u64')
    sk, order = torch.sort(k_dev, stable=True)
    assert torch.equal(ko, sk)
    if with_vals:
        assert torch.equal(vo, vals[order])
    assert torch.equal(k_dev, k_in)


@pytest.mark.parametrize('ks,stride', [(3, 1), (2, 2)])
def test_neighbour_table_from_torchsparse_rule_lists(ks, stride):
    """lidal_kmap_from_rules: a caller holding what backend.convolution_forward_cuda receives (nbmaps,
    nbsizes in torchsparse order) gets the table the convolution kernels consume -- equal to the one
    the library builds itself -- and out-of-range rules are counted, not written."""
    from lidal_amd import backend as B
    F = _F()
    c = _surface_coords(40, 2, seed=ks).to(DEV)
    km, oc = F.build_kernel_map(c, (1, 1, 1), (ks,) * 3, (stride,) * 3)
    k, n_out = km.nbr_out.shape
    maps = km.nbmaps.contiguous()
    got = torch.empty_like(km.nbr_out)
    bad = torch.empty(1, dtype=torch.int32, device=DEV)
    B.check(B.lib().lidal_kmap_from_rules(B.ptr(maps), B.ptr(km.nbsizes), k, maps.shape[0], c.shape[0], n_out,
                                          B.ptr(got), B.ptr(bad), B.stream()), 'kmap_from_rules')
    assert int(bad.item()) == 0 and torch.equal(got, km.nbr_out)
    # with spare capacity behind the rules (torchsparse's buffer) and one corrupt rule
    cap = torch.cat([maps, torch.full((7, 2), 123456789, dtype=torch.int, device=DEV)])
    cap[3, 0] = c.shape[0] + 5
    B.check(B.lib().lidal_kmap_from_rules(B.ptr(cap), B.ptr(km.nbsizes), k, cap.shape[0], c.shape[0], n_out,
                                          B.ptr(got), B.ptr(bad), B.stream()), 'kmap_from_rules')
    assert int(bad.item()) == 1
    # ... and the convolution runs from it
    order = F.conv.RowOrder(got)
    assert order.n_rows == n_out


@pytest.mark.parametrize('c,relu', [(32, True), (96, True), (128, False), (256, True)])
def test_data_gradient_launch_leaves_the_batch_norm_backward_sums(c, relu):
    """lidal_conv_dgrad_bn_sums + lidal_bn_bwd_tiles against the separate route (lidal_conv_apply_image, then
    lidal_bn_bwd with its own pass for the sums): the data gradient is bitwise the same, grad_gamma /
    grad_beta agree to 1e-5 of the largest (f32 per-tile sums merged in f64 against f64 throughout), dx
    to bf16 rounding."""
    from lidal_amd.nn.functional import conv as C
    from lidal_amd.nn.functional import norm as N
    F = _F()
    coords = _surface_coords(60, 2, seed=c).to(DEV)
    n = coords.shape[0]
    km, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    g = torch.Generator().manual_seed(c)
    w = (torch.randn(27, c, c, generator=g) * 0.05).to(DEV)
    x_bn = (torch.randn(n, c, generator=g) * 1.5 + 0.3).to(DEV).bfloat16()       # bn input (conv1 output)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.2).to(DEV)
    mean = x_bn.float().mean(0)
    invstd = torch.rsqrt(x_bn.float().var(0, unbiased=False) + 1e-5)
    gout = torch.randn(n, c, generator=g).to(DEV).bfloat16()
    order, kflip = C._bwd_order(km, False)
    img = C._weight_image(w, torch.bfloat16, n, 1)
    plain = C._apply(gout, img, 27, c, order, kflip)
    fused = C._apply(gout, img, 27, c, order, kflip, None, False, (x_bn, mean, invstd, gamma, beta, relu))
    assert torch.equal(plain, fused)
    sums = fused._lidal_bnb_sums
    assert sums.shape == (-(-n // 128), c, 2)
    dx0, gg0, gb0, _ = N.train_backward(x_bn, gamma, beta, mean, invstd, relu, plain, True)
    dx1, gg1, gb1, _ = N.train_backward(x_bn, gamma, beta, mean, invstd, relu, fused, True, None, sums)
    for a, b in ((gg1, gg0), (gb1, gb0)):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max(), ((a - b).abs().max(), b.abs().max())
    assert (dx1.float() - dx0.float()).abs().max() <= 2 ** -7 * dx0.float().abs().max()


@pytest.mark.parametrize('ts,levels', [(1, 4), (2, 3), (1, 1)])
def test_downsample_pyramid_equals_the_chained_downsamples(ts, levels):
    """F.downsample_pyramid: every coarser level from one sort of the input voxels == spdownsample chained
    `levels` times (same rows, same (batch, x, y, z) order), on a ragged batch with non-contiguous batch ids."""
    F = _F()
    c = _coords(40000, extent=300, batches=3, seed=ts, stride=ts).to(DEV)
    c[:, 3] = c[:, 3] * 5 + 2                                     # batch ids 2, 7, 12
    pyr = F.downsample_pyramid(c, levels, ts)
    cur, s = c, ts
    for l in range(levels):
        cur = F.spdownsample(cur, 2, 2, s)
        s *= 2
        assert torch.equal(pyr[l], cur), l
    assert len(F.downsample_pyramid(torch.zeros((0, 4), dtype=torch.int, device=DEV), 2, 1)[1]) == 0


@pytest.mark.parametrize('dtype,c', [(torch.float32, 4), (torch.float32, 32), (torch.bfloat16, 32), (torch.bfloat16, 256)])
def test_one_point_per_voxel_voxelize_is_the_mean_form(dtype, c):
    """When every voxel holds exactly one point (what initial_voxelize finds on LiDAL's pre-voxelised scans),
    F.spvoxelize moves rows (lidal_voxelize_fwd_1to1) instead of building contributor lists: bit-equal output
    and gradient to the general form."""
    F = _F()
    g = torch.Generator().manual_seed(c)
    n = 50001
    idx = torch.randperm(n, generator=g).to(DEV)
    counts = torch.ones(n, dtype=torch.int, device=DEV)
    x = torch.randn(n, c, generator=g).to(DEV).to(dtype)
    outs = []
    for fast in (False, True):
        i2 = idx.clone()
        if fast:
            i2._lidal_one_to_one = True
        xi = x.clone().requires_grad_(True)
        y = F.spvoxelize(xi, i2, counts)
        y.backward(torch.arange(n * c, device=DEV).reshape(n, c).to(dtype) * 1e-3)
        outs.append((y.detach(), xi.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[1][0][idx], x)


def test_revoxelize_coords_is_the_reference_expression_bit_for_bit():
    """lidal_revoxelize_coords == `cat([(C[:, :3] * init_res) / after_res, C[:, -1:]], 1)` and its floor().int()
    (network/utils.py:14-17) computed by torch on the CPU: separately rounded IEEE multiply and divide."""
    from lidal_amd import backend as B
    g = torch.Generator().manual_seed(21)
    n = 200001
    c = torch.cat([torch.randint(0, 8192, (n, 3), generator=g).float(), torch.randint(0, 5, (n, 1), generator=g).float()], 1)
    c[:1000, :3] += torch.rand(1000, 3, generator=g)            # not only integers
    for init_res, after_res in ((0.05, 0.05), (0.05, 0.1), (1.0, 3.0), (0.3, 0.05)):
        res = torch.full((), after_res, dtype=torch.float32)
        want = torch.cat([(c[:, :3] * init_res) / res, c[:, -1].view(-1, 1)], 1)
        cg = c.to(DEV)
        got_f = torch.empty_like(cg)
        got_i = torch.empty(cg.shape, dtype=torch.int32, device=DEV)
        B.check(B.lib().lidal_revoxelize_coords(B.ptr(cg), n, init_res, after_res, B.ptr(got_f), B.ptr(got_i), B.stream()),
                'revoxelize_coords')
        assert torch.equal(got_f.cpu(), want), (init_res, after_res)
        assert torch.equal(got_i.cpu(), torch.floor(want).int())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_with_split_offsets_equals_the_unsplit_kernel(dtype):
    """On the coarse levels (few 128-row tiles, long chains of (offset, slice) phases) lidal_conv_apply_image_ws
    splits each tile's active offsets over 2-4 workgroups and a second kernel adds the f32 partial tiles and runs the
    epilogue (csrc/conv_img.hip, Split).  Against the unsplit launch (no workspace): the f32 sums differ only by their
    association -- 2e-6 relative in f32, at most one flipped rounding per few thousand elements in bf16 -- and the
    epilogue products (BatchNorm tile statistics, residual + ReLU, the data gradient's BatchNorm sums) follow."""
    from lidal_amd import backend as B
    F = _F()
    L = B.lib()
    code = B.dtype_code(dtype)
    g = torch.Generator().manual_seed(31)
    coords = _surface_coords(40, 2, seed=9).to(DEV)                  # 3200 rows = 25 tiles: the layer shapes below split
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    order = kmap.order_out
    n = coords.shape[0]
    for ci, co in ((128, 128), (256, 256), (192, 128), (64, 64)):
        x = torch.randn(n, ci, generator=g).to(dtype).to(DEV)
        w = (torch.randn(27, ci, co, generator=g) * 0.1).to(DEV)
        nb = L.lidal_conv_weight_image_bytes(27, ci, co, code, n)
        img = torch.empty(nb, dtype=torch.uint8, device=DEV)
        B.check(L.lidal_conv_weight_image(B.ptr(w), B.F32, 0, B.ptr(img), code, 27, ci, co, n, B.stream()), 'image')
        wsb = L.lidal_conv_apply_workspace_bytes(n, co)
        assert wsb > 0
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        res = torch.randn(n, co, generator=g).to(dtype).to(DEV)
        tiles = -(-n // L.lidal_conv_stats_tile_rows())
        for kflip, with_res in ((0, False), (1, True)):
            outs, stats = [], []
            for use_ws in (False, True):
                out = torch.full((n, co), float('nan'), dtype=dtype, device=DEV)
                st = torch.zeros((tiles, co, 3), dtype=torch.float32, device=DEV) if not with_res else None
                B.check(L.lidal_conv_apply_image_ws(B.ptr(x), B.ptr(img), B.ptr(order.table), B.ptr(order.perm),
                                                    B.ptr(order.tile_masks), B.ptr(out), n, n, ci, co, 27, kflip, code,
                                                    None, None, 2 if with_res else 0, B.ptr(res) if with_res else None,
                                                    B.ptr(st), B.ptr(ws) if use_ws else None, wsb if use_ws else 0,
                                                    B.stream()), 'conv')
                outs.append(out.double())
                stats.append(st)
            a, b = outs
            assert not torch.isnan(b).any()
            scale = float(a.abs().max())
            if dtype == torch.float32:          # the f32 parity mode is never split: the workspace changes nothing
                assert torch.equal(a, b)
            else:                       # a flipped rounding here and there: one bf16 ulp of the value (or of a summand that cancelled)
                assert float((a != b).double().mean()) < 5e-3
                ulp = 2.0 ** -7 * a.abs().clamp_min(0.05 * scale)
                assert float(((a - b).abs() > ulp).double().mean()) < 1e-4          # (a flip in a summand that then cancelled)
                assert bool(((a - b).abs() <= 4 * ulp).all())
            if stats[0] is not None:
                assert torch.equal(stats[0][:, :, 0], stats[1][:, :, 0])                    # counts
                assert float((stats[1][:, :, 1] - stats[0][:, :, 1]).abs().max()) < 2e-3 * scale      # per-tile means
        if dtype == torch.bfloat16:                 # the data gradient that also leaves BatchNorm backward sums
            bx = torch.randn(n, co, generator=g).to(dtype).to(DEV)
            mean, invstd = torch.randn(co, generator=g).to(DEV) * 0.1, (torch.rand(co, generator=g) + 0.5).to(DEV)
            gam, bet = (torch.rand(co, generator=g) + 0.5).to(DEV), torch.randn(co, generator=g).to(DEV) * 0.1
            sums = []
            for use_ws in (False, True):
                out = torch.empty((n, co), dtype=dtype, device=DEV)
                sm = torch.zeros((tiles, co, 2), dtype=torch.float32, device=DEV)
                B.check(L.lidal_conv_dgrad_bn_sums_ws(B.ptr(x), B.ptr(img), B.ptr(order.table), B.ptr(order.perm),
                                                      B.ptr(order.tile_masks), B.ptr(out), n, n, ci, co, 27, 1, code,
                                                      B.ptr(bx), B.ptr(mean), B.ptr(invstd), B.ptr(gam), B.ptr(bet), 1,
                                                      B.ptr(sm), B.ptr(ws) if use_ws else None, wsb if use_ws else 0,
                                                      B.stream()), 'dgrad')
                sums.append(sm.double().sum(0))
            assert _relerr(sums[1], sums[0]) < 5e-3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c', [(396662, 96), (22013, 128), (1001, 256), (77, 32)])
def test_block_tail_mask_with_batchnorm_sums_is_the_two_separate_passes(dtype, n, c):
    """lidal_add_relu_bwd_bn_sums + lidal_bn_bwd_from_sums (the tail of a residual block backwards as the planned step
    runs it, network/plan.py b_res; network/utils.py:142-172 backwards) against lidal_add_relu_bwd + lidal_bn_bwd: the
    masked gradient, both BatchNorms' data gradients and parameter gradients BITWISE (the partial sums are taken over
    the same terms in the same order)."""
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.norm import train_backward
    L = B.lib()
    g = torch.Generator(device='cpu').manual_seed(n + c)
    dev = torch.device(DEV)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev)
    out = rnd(n, c).to(dtype)
    grad = rnd(n, c).to(dtype)
    xa, xb = (rnd(n, c) * 1.5 + 0.3).to(dtype), (rnd(n, c) * 0.7 - 0.2).to(dtype)
    code = B.dtype_code(dtype)
    stats = []
    for x in (xa, xb):
        xf = x.float()
        stats.append((xf.mean(0).contiguous(), (1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)).contiguous()))
    wa, ba, wb, bb = rnd(c), rnd(c), rnd(c), rnd(c)
    # the separate passes with f64 partial sums (the f32 mode's arithmetic; bf16 lidal_bn_bwd takes f32 slab sums by
    # default since round 5: test_batchnorm_backward_slab_sums_against_the_f64_sums)
    was = L.lidal_bn_set_slab_sums(0)
    try:
        dxa, gga, gba, gm = train_backward(xa, wa, ba, stats[0][0], stats[0][1], False, grad, mask_from=out)
        dxb, ggb, gbb, _ = train_backward(xb, wb, bb, stats[1][0], stats[1][1], False, gm)
    finally:
        L.lidal_bn_set_slab_sums(was)
    nb = L.lidal_bn_workspace_bytes(n, c)
    for dual in (True, False):
        gm2 = torch.empty_like(out)
        pa = torch.empty(nb, dtype=torch.uint8, device=dev)
        pb = torch.empty(nb, dtype=torch.uint8, device=dev)
        B.check(L.lidal_add_relu_bwd_bn_sums(B.ptr(out), B.ptr(grad), B.ptr(gm2), code, n, c, B.ptr(xa), B.ptr(stats[0][0]),
                                             B.ptr(stats[0][1]), B.ptr(pa), B.ptr(xb) if dual else None,
                                             B.ptr(stats[1][0]) if dual else None, B.ptr(stats[1][1]) if dual else None,
                                             B.ptr(pb) if dual else None, nb, B.stream()), 'add_relu_bwd')
        assert torch.equal(gm2, gm)
        for x, w, b, (mu, inv), part, want in ((xa, wa, ba, stats[0], pa, (dxa, gga, gba)),
                                               (xb, wb, bb, stats[1], pb, (dxb, ggb, gbb)))[:2 if dual else 1]:
            dx = torch.empty_like(x)
            gg = torch.empty(c, dtype=torch.float32, device=dev)
            gb = torch.empty(c, dtype=torch.float32, device=dev)
            B.check(L.lidal_bn_bwd_from_sums(B.ptr(x), B.ptr(gm2), c, code, n, c, B.ptr(w), B.ptr(b), 0, B.ptr(mu), B.ptr(inv),
                                             B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(part), nb, B.stream()), 'bn_bwd')
            assert torch.equal(dx, want[0]) and torch.equal(gg, want[1]) and torch.equal(gb, want[2])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c', [(396662, 96), (226469, 32), (105363, 128), (1001, 256), (77, 32)])
def test_block_tail_mask_with_slab_sums_against_the_separate_passes(dtype, n, c):
    """lidal_add_relu_bwd_bn_tile_sums + lidal_bn_bwd_tiles (the tail of a residual block backwards on the levels with
    many rows, network/plan.py b_res; network/utils.py:142-172 backwards) against lidal_add_relu_bwd + lidal_bn_bwd: the
    masked gradient BITWISE; the slab sums against f64 sums of the same rows (f32 sums of <= ~800 rows: 2e-6 of the
    column's sum of magnitudes); parameter gradients within 1e-5 of the columns' scale, dx within one rounding step of the
    element type -- the bars tests/test_ops_gpu.py holds the convolutions' tile sums to."""
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.norm import train_backward
    L = B.lib()
    g = torch.Generator(device='cpu').manual_seed(7 * n + c)
    dev = torch.device(DEV)

    def rnd(*shape):
        return torch.randn(*shape, generator=g).to(dev)
    out = rnd(n, c).to(dtype)
    grad = rnd(n, c).to(dtype)
    xa, xb = (rnd(n, c) * 1.5 + 0.3).to(dtype), (rnd(n, c) * 0.7 - 0.2).to(dtype)
    code = B.dtype_code(dtype)
    stats = []
    for x in (xa, xb):
        xf = x.float()
        stats.append((xf.mean(0).contiguous(), (1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)).contiguous()))
    wa, ba, wb, bb = rnd(c), rnd(c), rnd(c), rnd(c)
    dxa, gga, gba, gm = train_backward(xa, wa, ba, stats[0][0], stats[0][1], False, grad, mask_from=out)
    dxb, ggb, gbb, _ = train_backward(xb, wb, bb, stats[1][0], stats[1][1], False, gm)
    parts = int(L.lidal_bn_tail_parts(n, c, code))
    # (csrc/bn.hip rows_per_wg_ew: 512 slabs from 12 MiB of rows on, 256 below, at least 32 rows each)
    rpw = max(32, -(-n // (512 if n * c * out.element_size() >= (12 << 20) else 256)))
    assert parts == -(-n // rpw) and 1 <= parts <= 512
    for dual in (True, False):
        gm2 = torch.empty_like(out)
        sa = torch.full((c, parts, 2), float('nan'), dtype=torch.float32, device=dev)
        sb = torch.full((c, parts, 2), float('nan'), dtype=torch.float32, device=dev)
        B.check(L.lidal_add_relu_bwd_bn_tile_sums(B.ptr(out), B.ptr(grad), B.ptr(gm2), code, n, c, B.ptr(xa),
                                                  B.ptr(stats[0][0]), B.ptr(stats[0][1]), B.ptr(sa),
                                                  B.ptr(xb) if dual else None, B.ptr(stats[1][0]) if dual else None,
                                                  B.ptr(stats[1][1]) if dual else None, B.ptr(sb) if dual else None,
                                                  parts, B.stream()), 'add_relu_bwd')
        assert torch.equal(gm2, gm)
        for x, w, b, (mu, inv), sums, want in ((xa, wa, ba, stats[0], sa, (dxa, gga, gba)),
                                               (xb, wb, bb, stats[1], sb, (dxb, ggb, gbb)))[:2 if dual else 1]:
            # the sums of every slab against f64
            gd = gm.double()
            xh = (x.double() - mu.double()) * inv.double()
            pad = parts * rpw - n
            def slabs(t):
                t = torch.cat([t, torch.zeros(pad, c, dtype=t.dtype, device=dev)]) if pad else t
                return t.view(parts, rpw, c).sum(1).t()                  # [c, parts]
            ref_a, ref_b = slabs(gd), slabs(gd * xh)
            mag_a, mag_b = slabs(gd.abs()), slabs((gd * xh).abs())
            assert torch.isfinite(sums).all()
            assert float(((sums[:, :, 0].double() - ref_a).abs() / (mag_a + 1e-30)).max()) < 2e-6
            assert float(((sums[:, :, 1].double() - ref_b).abs() / (mag_b + 1e-30)).max()) < 2e-6
            dx = torch.empty_like(x)
            gg = torch.empty(c, dtype=torch.float32, device=dev)
            gb = torch.empty(c, dtype=torch.float32, device=dev)
            B.check(L.lidal_bn_bwd_tiles(B.ptr(x), B.ptr(gm2), c, code, n, c, B.ptr(w), B.ptr(b), 0, B.ptr(mu), B.ptr(inv),
                                         B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(sums), parts, B.stream()), 'bn_bwd')
            scale_g = float(mag_b.sum(1).max())
            scale_b = float(mag_a.sum(1).max())
            assert float((gg.double() - want[1].double()).abs().max()) < 1e-5 * scale_g
            assert float((gb.double() - want[2].double()).abs().max()) < 1e-5 * scale_b
            err = (dx.double() - want[0].double()).abs()
            # (f32 is not a product case -- norm.tail_tiles is bf16 only --: the sums' 1e-6 shows in dx there)
            step = (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -16) * want[0].double().abs().clamp_min(1e-3)
            assert float((err / step).max()) <= 1.0 + 1e-6


@pytest.mark.parametrize('relu', [0, 1])
@pytest.mark.parametrize('n,c,ld', [(396662, 96, 128), (396662, 256, 256), (105363, 128, 192), (43145, 256, 384), (1001, 64, 64), (77, 32, 32)])
def test_batchnorm_backward_slab_sums_against_the_f64_sums(n, c, ld, relu):
    """bf16 lidal_bn_bwd with its sums taken as f32 slab sums (the default since round 5: bn_bwd_slab_sums_kernel, merged
    in f64 like the convolutions' tile sums) against the f64 partial sums (lidal_bn_set_slab_sums(0)), on gradient slices
    of a wider matrix too (dy_stride > c: the decoder's concatenations): parameter gradients within 1e-5 of the column's
    sum of magnitudes, dx within one bf16 rounding step; and the fused launch bitwise the separate launches."""
    from lidal_amd import backend as B
    L = B.lib()
    dev = torch.device(DEV)
    g = torch.Generator(device='cpu').manual_seed(11 * n + c + relu)
    x = (torch.randn(n, c, generator=g) * 1.3 + 0.2).to(dev).bfloat16()
    wide = torch.randn(n, ld, generator=g).to(dev).bfloat16()
    dy = wide[:, ld - c:]
    w, b = torch.randn(c, generator=g).to(dev), torch.randn(c, generator=g).to(dev)
    xf = x.float()
    mu = xf.mean(0).contiguous()
    inv = (1.0 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)).contiguous()
    nb = L.lidal_bn_workspace_bytes(n, c)

    def run():
        dx = torch.empty_like(x)
        gg = torch.empty(c, dtype=torch.float32, device=dev)
        gb = torch.empty(c, dtype=torch.float32, device=dev)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        B.check(L.lidal_bn_bwd(B.ptr(x), dy.data_ptr(), ld, 1, n, c, B.ptr(w), B.ptr(b), relu, B.ptr(mu), B.ptr(inv),
                               B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(ws), nb, B.stream()), 'bn_bwd')
        torch.cuda.synchronize()
        return dx, gg, gb
    was = L.lidal_bn_set_slab_sums(0)
    try:
        ref = run()
        L.lidal_bn_set_slab_sums(1)
        got = run()
        L.lidal_bn_set_fused(0)
        apart = run()
    finally:
        L.lidal_bn_set_fused(1)
        L.lidal_bn_set_slab_sums(was)
    assert all(torch.equal(a, b_) for a, b_ in zip(got, apart))
    xh = (x.double() - mu.double()) * inv.double()
    d = dy.double()
    if relu:
        d = torch.where(xh * w.double() + b.double() > 0, d, torch.zeros_like(d))
    scale_b, scale_g = float(d.abs().sum(0).max()), float((d * xh).abs().sum(0).max())
    assert float((got[2].double() - ref[2].double()).abs().max()) < 1e-5 * scale_b
    assert float((got[1].double() - ref[1].double()).abs().max()) < 1e-5 * scale_g
    err = (got[0].double() - ref[0].double()).abs()
    assert float((err / (2.0 ** -7 * ref[0].double().abs().clamp_min(1e-3))).max()) <= 1.0 + 1e-6


def _tile_triples(x, tile=128):
    """(count, mean, M2) per 128-row tile and channel, what a convolution's epilogue leaves: f32, laid out
    [c][tiles][3] (csrc/conv_img.hip store_tile) -- returned with that buffer's shape recorded as (tiles, c, 3)."""
    n, c = x.shape
    xf = x.float()
    out = []
    for r in range(0, n, tile):
        blk = xf[r:r + tile]
        m = blk.mean(0)
        out.append(torch.stack([torch.full_like(m, blk.shape[0]), m, ((blk - m) ** 2).sum(0)], 1))
    t = torch.stack(out, 0)                                     # [tiles, c, 3]
    return t.permute(1, 0, 2).contiguous().view(t.shape[0], t.shape[1], 3)     # the bytes of [c][tiles][3]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c', [(396662, 96), (83177, 32), (22013, 128), (3300, 256), (77, 64)])
def test_batchnorm_merges_inside_their_consumers_are_the_separate_launches(dtype, n, c):
    """lidal_bn_set_fused: the tile-statistics merge inside the apply launch (forward) and the partial-sum merges inside
    the dx launch (backward) against the separate merge launches: outputs, saved statistics, running statistics and
    parameter gradients BITWISE, with and without the residual / ReLU epilogues, on sizes from one workgroup to 512."""
    from lidal_amd import backend as B
    L = B.lib()
    dev = torch.device(DEV)
    g = torch.Generator(device='cpu').manual_seed(3 * n + c)
    x = (torch.randn(n, c, generator=g) * 1.7 + 0.4).to(dev).to(dtype)
    res = torch.randn(n, c, generator=g).to(dev).to(dtype)
    dy = torch.randn(n, c, generator=g).to(dev).to(dtype)
    gamma, beta = torch.randn(c, generator=g).to(dev), torch.randn(c, generator=g).to(dev)
    tiles = _tile_triples(x)
    tile_sums = torch.randn(tiles.shape[0], c, 2, generator=g).to(dev).contiguous()
    code = B.dtype_code(dtype)
    nb = L.lidal_bn_workspace_bytes(n, c)
    outs = {}
    try:
        for fused in (0, 1):
            L.lidal_bn_set_fused(fused)
            got = []
            for relu, r in ((1, None), (3, res), (0, None)):
                y = torch.empty_like(x)
                mean = torch.empty(c, device=dev)
                inv = torch.empty(c, device=dev)
                rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
                nbt = torch.zeros(1, dtype=torch.int64, device=dev)
                B.check(L.lidal_bn_train_fwd_tiles(B.ptr(x), code, n, c, B.ptr(gamma), B.ptr(beta), 1e-5, 0.1, B.ptr(rm), B.ptr(rv),
                                                   B.ptr(nbt), relu, B.ptr(r), B.ptr(y), B.ptr(mean), B.ptr(inv), B.ptr(tiles),
                                                   tiles.shape[0], B.stream()), 'bn_train_fwd')
                got += [y, mean, inv, rm, rv, nbt]
                # backward through lidal_bn_bwd (partial + merge + dx)
                dx = torch.empty_like(x)
                gg, gb = torch.empty(c, device=dev), torch.empty(c, device=dev)
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                B.check(L.lidal_bn_bwd(B.ptr(x), B.ptr(dy), c, code, n, c, B.ptr(gamma), B.ptr(beta), relu & 1, B.ptr(mean), B.ptr(inv),
                                       B.ptr(dx), B.ptr(gg), B.ptr(gb), B.ptr(ws), nb, B.stream()), 'bn_bwd')
                got += [dx, gg, gb]
                # ... and through lidal_bn_bwd_tiles (f32 sums per 128-row tile, as a data-gradient launch leaves them)
                dx2 = torch.empty_like(x)
                gg2, gb2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
                B.check(L.lidal_bn_bwd_tiles(B.ptr(x), B.ptr(dy), c, code, n, c, B.ptr(gamma), B.ptr(beta), relu & 1, B.ptr(mean),
                                             B.ptr(inv), B.ptr(dx2), B.ptr(gg2), B.ptr(gb2), B.ptr(tile_sums), tile_sums.shape[0],
                                             B.stream()), 'bn_bwd')
                got += [dx2, gg2, gb2]
            torch.cuda.synchronize()
            outs[fused] = got
    finally:
        L.lidal_bn_set_fused(1)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[1][0].float()).all()


@pytest.mark.parametrize('ci,co', [(32, 32), (96, 96), (128, 64), (64, 20)])
def test_split_f32_convolution_against_f64(ci, co):
    """LIDAL_F32_SPLIT (csrc/conv_img.hip conv_split_kernel): f32 features and weights, every operand cut into three bf16
    pieces, six partial products on the bf16 MFMA, f32 accumulation.  Against an f64 product on the same rule list the
    result must be as close as the exact-f32 kernel's (both within 5e-6 of the output scale: the error of an f32 dot
    product of ~27 * ci terms), the epilogue (affine map, ReLU, residual) and the dense form (identity rule list) too."""
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.conv import _weight_image
    F = _F()
    coords = _surface_coords(70, 2, seed=ci + co).to(DEV)
    n = coords.shape[0]
    km, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    o = km.order_out
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = (torch.randn(n, ci, generator=g) * 2 + 0.1).to(DEV)
    w = (torch.randn(27, ci, co, generator=g) * 0.1).to(DEV)
    scale = (torch.rand(co, generator=g) + 0.5).to(DEV)
    shift = torch.randn(co, generator=g).to(DEV)
    res = torch.randn(n, co, generator=g).to(DEV)
    nbr = km.nbr_out.long()
    ref = torch.zeros(n, co, dtype=torch.float64, device=DEV)
    for k in range(27):
        m = nbr[k] >= 0
        ref[m] += x.double()[nbr[k][m]] @ w.double()[k]
    ref_ep = torch.relu(ref * scale.double() + shift.double()) + res.double()
    L = B.lib()
    errs = {}
    for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
        with torch.no_grad():
            img = _weight_image(w, torch.float32, n, 0, code)
        y = torch.empty((n, co), dtype=torch.float32, device=DEV)
        B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks), B.ptr(y),
                                         n, n, ci, co, 27, 0, code, None, None, 0, None, None, B.stream()), 'conv')
        y2 = torch.empty((n, co), dtype=torch.float32, device=DEV)
        B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks), B.ptr(y2),
                                         n, n, ci, co, 27, 0, code, B.ptr(scale), B.ptr(shift), 1, B.ptr(res), None,
                                         B.stream()), 'conv')
        errs[name] = (float((y.double() - ref).abs().max() / ref.abs().max()),
                      float((y2.double() - ref_ep).abs().max() / ref_ep.abs().max()))
    assert errs['split'][0] < 5e-6 and errs['split'][1] < 5e-6, errs
    assert errs['split'][0] < 4 * errs['exact'][0] + 1e-7, errs
    # dense form (k = 1, no table)
    wd = (torch.randn(1, ci, co, generator=g) * 0.1).to(DEV)
    refd = x.double() @ wd.double()[0]
    with torch.no_grad():
        img = _weight_image(wd, torch.float32, n, 0, B.F32_SPLIT)
    y = torch.empty((n, co), dtype=torch.float32, device=DEV)
    B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), None, None, None, B.ptr(y), n, n, ci, co, 1, 0, B.F32_SPLIT, None,
                                     None, 0, None, None, B.stream()), 'conv')
    assert float((y.double() - refd).abs().max() / refd.abs().max()) < 2e-6
    # an operand whose pieces matter: values with all 24 significand bits set give the exact product of small integers
    xi = torch.full((n, ci), 1.0 + 2.0 ** -23, device=DEV)
    wi = torch.full((1, ci, co), 1.0 - 2.0 ** -24, device=DEV)
    with torch.no_grad():
        img = _weight_image(wi, torch.float32, n, 0, B.F32_SPLIT)
    B.check(L.lidal_conv_apply_image(B.ptr(xi), B.ptr(img), None, None, None, B.ptr(y), n, n, ci, co, 1, 0, B.F32_SPLIT, None,
                                     None, 0, None, None, B.stream()), 'conv')
    want = ci * (1.0 + 2.0 ** -23) * (1.0 - 2.0 ** -24)
    assert float((y.double() - want).abs().max()) <= ci * 2.0 ** -22, float((y.double() - want).abs().max())


def test_f32_inference_runs_in_the_split_form_and_training_does_not():
    """backend.conv_code: under no_grad an f32 network's convolutions and dense layers take LIDAL_F32_SPLIT (except the
    4-channel stem).  With autograd on (the f32 training step) -- round 6 -- the SPARSE convolutions whose two channel
    counts are whole 32-channel slices take it too, forward and data gradient, and every weight gradient whose channel
    counts are whole 16-byte segments (lidal_conv_wgrad with LIDAL_F32_SPLIT: all but the 4-channel stem's); the dense
    layers' products and the 4-channel stem stay on the exact f32 MFMA.  LIDAL_F32_SPLIT_TRAIN=0 (backend.SPLIT_F32_TRAIN)
    keeps all of the training step there."""
    import lidal_amd
    from lidal_amd import backend as B
    assert B.conv_code(torch.float32, 96, True) == B.F32_SPLIT
    assert B.conv_code(torch.float32, 4, True) == B.F32 and B.conv_code(torch.float32, 96, False) == B.F32
    assert B.conv_code(torch.float32, 96, False, 96) == B.F32_SPLIT and B.conv_code(torch.float32, 96, False, 19) == B.F32
    assert B.conv_code(torch.float32, 4, False, 32) == B.F32
    assert B.conv_code(torch.bfloat16, 96, True) == B.BF16 and B.conv_code(torch.bfloat16, 96, False, 96) == B.BF16
    seen = []
    B.set_call_timer(lambda name, a, e0, e1: seen.append((name, [getattr(v, 'value', v) for v in a])))
    try:
        from lidal_amd.network import MinkUNet, plan
        from lidal_amd import synth
        b = synth.make_train_batch(n_frames=1, n_points=6000, seed=3)
        feats, coords = (torch.from_numpy(b[k]).to(DEV) for k in ('feats_v_b', 'coords_v_b'))
        model = MinkUNet(19).to(DEV).eval()
        saved = plan.ENABLED
        plan.ENABLED = False
        try:
            with torch.no_grad():
                model(lidal_amd.SparseTensor(feats, coords))
            infer = [a[12] for nme, a in seen if nme in ('lidal_conv_apply_image', 'lidal_conv_apply_image_ws')]
            model.train()
            trains = {}
            for split_train in (True, False):
                del seen[:]
                B.SPLIT_F32_TRAIN = split_train
                model(lidal_amd.SparseTensor(feats, coords))[0].sum().backward()
                # (k, dtype code) of every forward product / data gradient, and the dtype code of every weight gradient
                trains[split_train] = ([(a[10], a[12]) for nme, a in seen if nme in ('lidal_conv_apply_image', 'lidal_conv_apply_image_ws')],
                                       [a[13] for nme, a in seen if nme == 'lidal_conv_wgrad'])
        finally:
            plan.ENABLED = saved
            B.SPLIT_F32_TRAIN = True
    finally:
        B.set_call_timer(None)
    assert infer.count(B.F32_SPLIT) >= 40 and infer.count(B.F32) == 1, infer       # (the stem's first convolution: 4 channels)
    apply_on, wgrad_on = trains[True]
    assert [c for k, c in apply_on if k > 1].count(B.F32_SPLIT) >= 70                 # sparse convolutions, both directions
    assert all(c == B.F32 for k, c in apply_on if k == 1) and any(k == 1 for k, c in apply_on)      # dense layers: exact
    assert sum(1 for k, c in apply_on if k > 1 and c == B.F32) == 1                   # the 4-channel stem (no data gradient)
    assert wgrad_on.count(B.F32_SPLIT) >= 40 and wgrad_on.count(B.F32) == 2, wgrad_on       # (exact: the 4-channel stem, the 19-class head)
    assert B.wgrad_code(torch.float32, 96, 96) == B.F32_SPLIT and B.wgrad_code(torch.float32, 4, 32) == B.F32
    assert B.wgrad_code(torch.bfloat16, 96, 96) == B.BF16
    apply_off, wgrad_off = trains[False]
    assert apply_off and {c for k, c in apply_off} == {B.F32} and set(wgrad_off) == {B.F32}


@pytest.mark.parametrize('ci,co', [(32, 32), (96, 96), (128, 96), (256, 128), (64, 24)])
def test_split_f32_weight_gradient_against_f64(ci, co):
    """lidal_conv_wgrad with LIDAL_F32_SPLIT (csrc/wgrad_dma.hip wgrad_split_kernel, round 6: the f32 training step's weight
    gradients): f32 operands cut into three exact bf16 pieces by the call, six bf16 MFMAs per product, f32 accumulation.
    Against the f64 gradient it is as close as the exact f32 MFMA kernel (both within 4e-6 of the gradient's scale), bitwise
    reproducible, for sparse rule lists and for the dense (identity) form."""
    from lidal_amd import backend as B
    from lidal_amd import synth
    F = _F()
    L = B.lib()
    dev = torch.device(DEV)
    b = synth.make_train_batch(n_frames=1, n_points=30000, seed=17)
    coords = torch.from_numpy(b['coords_v_b']).to(dev)
    with torch.enable_grad():
        kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    n = coords.shape[0]
    g = torch.Generator(device='cpu').manual_seed(ci * 100 + co)
    x = torch.randn(n, ci, generator=g).to(dev)
    gy = (torch.randn(n, co, generator=g) * 0.3).to(dev)
    nbmaps, koff = kmap.nbmaps.long(), kmap.koff.cpu().tolist()
    ref = torch.zeros(27, ci, co, dtype=torch.float64, device=dev)
    for k in range(27):
        pr = nbmaps[koff[k]:koff[k + 1]]
        if pr.shape[0]:
            ref[k] = x.double()[pr[:, 0]].t() @ gy.double()[pr[:, 1]]
    scale = float(ref.abs().max())
    errs = {}
    for name, code in (('exact', B.F32), ('split', B.F32_SPLIT)):
        slabs = int(L.lidal_conv_wgrad_slabs(n, n, 27, ci, co, code))
        assert slabs > 0
        partial = torch.full((slabs, ci, co), float('nan'), dtype=torch.float32, device=dev)
        outs = []
        for _ in range(2):
            gw = torch.full((27, ci, co), float('nan'), dtype=torch.float32, device=dev)
            B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(kmap._nbmaps_cap), B.ptr(kmap.koff), 0, B.ptr(gw),
                                       B.ptr(partial), slabs, 27, ci, co, code, B.stream()), 'wgrad')
            outs.append(gw)
        assert torch.equal(outs[0], outs[1])
        errs[name] = float((outs[0].double() - ref).abs().max()) / scale
    assert errs['split'] < 4e-6 and errs['exact'] < 4e-6, errs
    # dense form: x^T gy
    koff1 = torch.tensor([0, n], dtype=torch.int64, device=dev)
    slabs = int(L.lidal_conv_wgrad_slabs(n, n, 1, ci, co, B.F32_SPLIT))
    partial = torch.empty((slabs, ci, co), dtype=torch.float32, device=dev)
    gw = torch.empty((1, ci, co), dtype=torch.float32, device=dev)
    B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, None, B.ptr(koff1), 0, B.ptr(gw), B.ptr(partial), slabs, 1, ci, co,
                               B.F32_SPLIT, B.stream()), 'wgrad(dense)')
    refd = x.double().t() @ gy.double()
    assert float((gw[0].double() - refd).abs().max()) < 4e-6 * float(refd.abs().max())
    # a shape it does not serve says so
    assert int(L.lidal_conv_wgrad_slabs(n, n, 27, 4, 32, B.F32_SPLIT)) == -1
