"""The streamed weight gradient (round 6): lidal_wgrad_streams_build + lidal_conv_wgrad_streams through the C-ABI.

torchsparse's convolution_backward_cuda (v1.4.0) fixes the result -- gw[k] = a[in_k]^T b[out_k] over the rule lists; the
streamed form computes that sum in another order (one rule stream per workgroup, csrc/wgrad_streams.hip).  Held here:
  * the device builder == its CPU restatement (oracle/streams_ref.py) BIT FOR BIT, keyed by row index and by a parent table;
  * the product within f32 rounding of the float64 sum (tolerance 2e-6 of the gradient's scale, as the offset-major
    kernel's own test), equal in that bound to lidal_conv_wgrad, bitwise reproducible, for every tile shape the step uses,
    for the channel-padded stem and for tiny / empty maps;
  * the planned step == the per-operator step BITWISE with the streams forced on at test size, both networks."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _level(points=30000, frames=2, seed=31, level=0):
    """A real kernel map: SPVCNN's geometry of a small synthetic batch (level 0 is numbered by coordinate hash)."""
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, Geometry
    b = synth.make_train_batch(n_frames=frames, n_points=points, seed=seed)
    coords = torch.from_numpy(b['coords_v_b']).to(DEV)
    model = SPVCNN(19).to(DEV).train()
    g = Geometry.build(model, coords, True)
    s = 1 << level
    km = g.x0.kmaps[((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    k2 = g.x0.kmaps[((s, s, s), (2, 2, 2), (2, 2, 2), (1, 1, 1))]
    return g, km, k2


def _build(km, key_tab, key_k, key_range):
    from lidal_amd import backend as B
    L = B.lib()
    k, n = km.nbr_out.shape
    n_wg = int(L.lidal_wgrad_streams_workgroups())
    cap = int(L.lidal_wgrad_streams_rules(n, k, n_wg))
    sp = torch.full((cap, 2), -7, dtype=torch.int32, device=DEV)
    sd = torch.full((int(L.lidal_wgrad_streams_desc_words(k, n_wg)),), -7, dtype=torch.int32, device=DEV)
    wsb = int(L.lidal_wgrad_streams_workspace_bytes(n, k))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    B.check(L.lidal_wgrad_streams_build(B.ptr(km._nbmaps_cap), B.ptr(km.koff), k, n, B.ptr(key_tab), key_k, key_range, n_wg,
                                        B.ptr(sp), cap, B.ptr(sd), B.ptr(ws), wsb, B.stream()), 'wgrad_streams_build')
    torch.cuda.synchronize()
    return sp, sd, n_wg


@pytest.mark.parametrize('keyed', [False, True])
def test_device_builder_is_the_restatement_bit_for_bit(keyed):
    from oracle import streams_ref as S
    g, km, k2 = _level()
    n = km.sizes[0]
    sizes = [int(v) for v in km.nbsizes.tolist()]
    key_tab, key_k, key_range, key = (None, 0, n, None)
    if keyed:                   # the parent row on level 1 (the strided map's inverse neighbour table)
        key_tab, key_k, key_range = k2.nbr_in, k2.volume, k2.sizes[1]
        key = k2.nbr_in.max(0)[0].cpu().numpy()
        assert key.min() >= 0
    sp, sd, n_wg = _build(km, key_tab, key_k, key_range)
    ref_sp, ref_sd = S.build_streams(km.nbmaps.cpu().numpy(), sizes, key, key_range, n, n_wg)
    assert np.array_equal(sd.cpu().numpy(), ref_sd)
    total = ref_sp.shape[0]
    assert np.array_equal(sp[:total].cpu().numpy(), ref_sp)
    assert bool((sp[total:] == -7).all())                   # nothing is written past the last stage
    # twice the same tables
    sp2, sd2, _ = _build(km, key_tab, key_k, key_range)
    assert torch.equal(sp2[:total], sp[:total]) and torch.equal(sd2, sd)


def _f64(x, g, km):
    ref = torch.zeros((km.volume, x.shape[1], g.shape[1]), dtype=torch.float64, device=DEV)
    xd, gd = x.double(), g.double()
    o = 0
    for kk, sz in enumerate(km.nbsizes.tolist()):
        pr = km.nbmaps[o:o + sz].long()
        ref[kk] = xd[pr[:, 0]].t() @ gd[pr[:, 1]]
        o += sz
    return ref


@pytest.mark.parametrize('ca,cb', [(32, 32), (96, 96), (128, 96), (64, 128), (128, 128), (8, 32), (96, 24)])
def test_streamed_product_against_f64_and_the_offset_major_kernel(ca, cb):
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.conv import wgrad_scratch
    g, km, k2 = _level()
    n = km.sizes[0]
    L = B.lib()
    assert L.lidal_conv_wgrad_streams_serves(n, n, 27, ca, cb) == 1
    sp, sd, n_wg = _build(km, k2.nbr_in, k2.volume, k2.sizes[1])
    torch.manual_seed(ca * 1000 + cb)
    x = torch.randn(n, ca, device=DEV).bfloat16()
    gy = torch.randn(n, cb, device=DEV).bfloat16()
    out = []
    for rep in range(2):
        gw = torch.full((27, ca, cb), float('nan'), dtype=torch.float32, device=DEV)
        partial = torch.full((2 * n_wg, ca, cb), float('nan'), dtype=torch.float32, device=DEV)     # (slabs that are read are written)
        B.check(L.lidal_conv_wgrad_streams(B.ptr(x), B.ptr(gy), n, n, B.ptr(sp), B.ptr(sd), n_wg, 0, B.ptr(gw), B.ptr(partial),
                                           partial.shape[0], 27, ca, cb, B.BF16, B.stream()), 'conv_wgrad_streams')
        out.append(gw)
    old = torch.empty((27, ca, cb), dtype=torch.float32, device=DEV)
    partial = wgrad_scratch(n, n, 27, ca, cb, torch.bfloat16, DEV)
    B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(km._nbmaps_cap), B.ptr(km.koff), 0, B.ptr(old), B.ptr(partial),
                               partial.shape[0], 27, ca, cb, B.BF16, B.stream()), 'conv_wgrad')
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1])                      # fixed slab order, no atomics
    ref = _f64(x, gy, km)
    scale = float(ref.abs().max())
    assert float((out[0].double() - ref).abs().max()) <= 2e-6 * scale
    assert float((old.double() - ref).abs().max()) <= 2e-6 * scale
    assert float((out[0] - old).abs().max()) <= 4e-6 * scale


def test_streams_refuse_what_they_do_not_serve():
    from lidal_amd import backend as B
    L = B.lib()
    assert L.lidal_conv_wgrad_streams_serves(1000, 1000, 27, 192, 128) == 0         # two channel tiles
    assert L.lidal_conv_wgrad_streams_serves(1000, 1000, 27, 96, 19) == 0           # not whole 16-byte segments
    g, km, k2 = _level(points=4000, frames=1)
    n = km.sizes[0]
    sp, sd, n_wg = _build(km, None, 0, n)
    x = torch.randn(n, 192, device=DEV).bfloat16()
    gy = torch.randn(n, 128, device=DEV).bfloat16()
    gw = torch.empty((27, 192, 128), dtype=torch.float32, device=DEV)
    partial = torch.empty((2 * n_wg, 192, 128), dtype=torch.float32, device=DEV)
    rc = L.lidal_conv_wgrad_streams(B.ptr(x), B.ptr(gy), n, n, B.ptr(sp), B.ptr(sd), n_wg, 0, B.ptr(gw), B.ptr(partial),
                                    partial.shape[0], 27, 192, 128, B.BF16, B.stream())
    assert rc != 0 and b'channel tile' in L.lidal_last_error()
    rc = L.lidal_conv_wgrad_streams(B.ptr(x), B.ptr(gy), n, n, B.ptr(sp), B.ptr(sd), n_wg, 0, B.ptr(gw), B.ptr(partial),
                                    n_wg, 27, 96, 96, B.BF16, B.stream())
    assert rc != 0 and b'slabs' in L.lidal_last_error()


def test_tiny_map_with_empty_offsets_and_idle_workgroups():
    """37 rows: most (XCD, offset) lists are empty, most workgroups idle; the reducer still writes every offset."""
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.conv import KernelMap
    from oracle import streams_ref as S
    rng = np.random.default_rng(3)
    n = 37
    nbr = np.full((27, n), -1, np.int32)
    nbr[13] = np.arange(n)
    for k in (0, 5, 26):
        rows = rng.permutation(n)[:rng.integers(1, 9)]
        nbr[k, rows] = rng.integers(0, n, len(rows))
    km = KernelMap(torch.from_numpy(nbr).to(DEV), (n, n), 27, True)
    sp, sd, n_wg = _build(km, None, 0, n)
    ref_sp, ref_sd = S.build_streams(km.nbmaps.cpu().numpy(), [int(v) for v in km.nbsizes.tolist()], None, n, n, n_wg)
    assert np.array_equal(sd.cpu().numpy(), ref_sd) and np.array_equal(sp[:ref_sp.shape[0]].cpu().numpy(), ref_sp)
    L = B.lib()
    x = torch.randn(n, 32, device=DEV).bfloat16()
    gy = torch.randn(n, 32, device=DEV).bfloat16()
    gw = torch.full((27, 32, 32), float('nan'), dtype=torch.float32, device=DEV)
    partial = torch.full((2 * n_wg, 32, 32), float('nan'), dtype=torch.float32, device=DEV)
    B.check(L.lidal_conv_wgrad_streams(B.ptr(x), B.ptr(gy), n, n, B.ptr(sp), B.ptr(sd), n_wg, 0, B.ptr(gw), B.ptr(partial),
                                       partial.shape[0], 27, 32, 32, B.BF16, B.stream()), 'conv_wgrad_streams')
    ref = _f64(x, gy, km)
    assert float((gw.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert bool((gw[1] == 0).all())                          # an offset without rules


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_planned_step_is_the_per_operator_step_with_the_streams_on(name, monkeypatch):
    """Both paths take the streamed form on every level the threshold admits (forced down to test size): the same entry
    point with the same arguments, so loss, logits and every gradient are bitwise equal; and the streamed launches
    really ran, in line and with the tables built ahead on the second stream."""
    from test_plan_gpu import _batches, _models, _steps
    from lidal_amd import backend as B
    monkeypatch.setattr(B, 'WGRAD_STREAMS_ROWS', 3000)
    torch.manual_seed(0)
    a = _models()[name](19).to(DEV).train()
    b = copy.deepcopy(a)
    c = copy.deepcopy(a)
    batches = _batches(3)
    la, ya, ga, _, ca = _steps(a, batches, True, planned=False, steps_with_grads=(0, 2))
    lb, yb, gb, _, cb = _steps(b, batches, True, planned=True, steps_with_grads=(0, 2))
    lc, yc, gc, _, cc = _steps(c, batches, True, planned=True, prefetch=True, steps_with_grads=(0, 2))
    assert ca.get('conv_wgrad_streams', 0) >= 8, ca         # (the per-operator path counts its library calls)
    assert la == lb == lc and torch.equal(ya, yb) and torch.equal(ya, yc)
    names = [k for k, _ in a.named_parameters()]
    for step in (0, 2):
        for k, p, q, r in zip(names, ga[step], gb[step], gc[step]):
            assert torch.equal(p, q) and torch.equal(p, r), (step, k)
    # the plans hold the same launches (tallied from the plan's words)
    from lidal_amd.network import plan
    from lidal_amd.train_step import forward_backward
    B.HITS.clear()
    saved = plan.TALLY
    plan.TALLY = True
    try:
        forward_backward(b, *batches[2], autocast=True)
    finally:
        plan.TALLY = saved
    assert B.HITS.get('conv_wgrad_streams', 0) == ca['conv_wgrad_streams'], (B.HITS, ca)
    # ... and against the step without them: the same sums in another order
    monkeypatch.setattr(B, 'WGRAD_STREAMS_ROWS', 0)
    torch.manual_seed(0)
    d = _models()[name](19).to(DEV).train()
    ld, yd, gd, _, cd = _steps(d, batches[:1], True, planned=True, steps_with_grads=(0,))
    assert 'conv_wgrad_streams' not in cd
    assert ld[0] == la[0]                                    # (the forward pass does not depend on the form)
    for k, p, q in zip(names, ga[0], gd[0]):
        sc = float(q.float().abs().max()) + 1e-12
        assert float((p.float() - q.float()).abs().max()) <= 1e-3 * sc, k
