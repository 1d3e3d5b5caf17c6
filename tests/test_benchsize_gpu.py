"""Parity AT the benchmarked size (BASELINE.json configs[1..4]): what bench.py times is what these
tests check.

  (a) one ~120 k-point scan (~83 k voxels) through a whole TRAIN step (train.py:127-140), SPVCNN
      and MinkUNet: f32 on the HIP path against the CPU oracle run in f64 (loss / logits 1e-4;
      sampled gradient norms within 2x the f32 oracle's own deviation from its f64 run, measured
      here at this size), then the bench dtype (bf16 autocast): cosine >= 0.99 on every sampled
      parameter.
  (b) the weight-gradient plan and the flipped-offset data gradient bench.py runs (5 scans,
      ~4e5 rows: W = 256 resident workgroups, 64-rule stages crossing offset boundaries, 32-bit
      byte offsets near 76 MB) through the C-ABI against f64 products computed by torch on the GPU
      from the same bf16 operands.
  (c) the inter-frame scorer on 120 k-point frames (NN grid of 120 k points, p x n_nei match
      threads, 63-bit cell-key sort), neighbour windows 10 and 24, against
      oracle.scoring_ref.score_frame (LiDAL.py:59-98).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

GKEYS = ['stem.0.kernel', 'stage2.1.net.0.kernel', 'stage4.2.net.3.kernel', 'up1.0.net.0.kernel',
         'up4.1.1.net.3.kernel', 'up4.1.0.net.0.kernel', 'classifier.0.weight', 'stage1.0.net.1.weight',
         'up3.1.0.downsample.0.kernel']


@pytest.fixture(scope='module')
def one_scan():
    from lidal_amd import synth
    b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
    return (torch.from_numpy(b['coords_v_b']), torch.from_numpy(b['feats_v_b']),
            torch.from_numpy(b['labels_v_b']))


@pytest.fixture(scope='module')
def bench_batch():
    from lidal_amd import synth
    b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    return torch.from_numpy(b['coords_v_b']).to(DEV)


def _oracle_step(cls, dt, coords, feats, labels, keys):
    from oracle import tsref
    from weights import fill_state_dict
    model = fill_state_dict(cls(19)).to(dt).train()
    if hasattr(model, 'dropout'):
        model.dropout.p = 0.0
    logits, _ = model(tsref.SparseTensor(feats.clone().to(dt), coords.clone()))
    loss = torch.nn.functional.cross_entropy(logits, labels, ignore_index=255, reduction='mean')
    loss.backward()
    named = dict(model.named_parameters())
    return loss.item(), logits.detach(), {k: named[k].grad.detach().double() for k in keys if k in named}


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_train_step_at_bench_size_matches_oracle(name, one_scan):
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.train_step import forward_backward
    from oracle.models_ref import MinkUNetRef, SPVCNNRef
    from weights import fill_state_dict
    coords, feats, labels = one_scan
    assert coords.shape[0] > 70000
    torch.set_num_threads(min(32, max(torch.get_num_threads(), (__import__('os').cpu_count() or 8))))
    ref_cls = {'spvcnn': SPVCNNRef, 'minkunet': MinkUNetRef}[name]
    keys = GKEYS + (['point_transforms.1.0.weight'] if name == 'spvcnn' else [])
    loss64, logits64, g64 = _oracle_step(ref_cls, torch.float64, coords, feats, labels, keys)
    # The yardstick for f32 gradients is the f32 CPU oracle's OWN distance from its f64 run -- and that
    # distance depends on the summation order: measured on this scan (MinkUNet) the oracle misses the f64
    # |g| of the stem by 5.7e-6 / 4.2e-5 / 2.0e-4 and of a BatchNorm gamma by 1.1e-4 / 3.0e-4 / 1.2e-3 with
    # 1 / 8 / 3 threads, and by other amounts again with the rows permuted (the f64 run moves by 1e-15):
    # the randomly initialised 49-layer net amplifies f32 rounding by ~10^4.  So three f32 oracle runs
    # with different summation orders (all cores, 3 threads, rows permuted) calibrate the bar.
    n_threads = torch.get_num_threads()
    f32_runs = [_oracle_step(ref_cls, torch.float32, coords, feats, labels, keys)[2]]
    torch.set_num_threads(3)
    f32_runs.append(_oracle_step(ref_cls, torch.float32, coords, feats, labels, keys)[2])
    torch.set_num_threads(n_threads)
    perm = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(1))
    f32_runs.append(_oracle_step(ref_cls, torch.float32, coords[perm], feats[perm], labels[perm], keys)[2])

    no_stats = [False]

    def run(autocast):
        model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).to(DEV).train()
        if hasattr(model, 'dropout'):
            model.dropout.p = 0.0
        if no_stats[0]:         # no BatchNorm statistics from the producing kernels
            n_off = 0
            for m in model.modules():
                if m.__dict__.get('bn_follows'):
                    m.bn_follows = False
                    n_off += 1
            assert n_off >= 49
        loss, logits = forward_backward(model, feats.to(DEV), coords.to(DEV), labels.to(DEV), autocast=autocast)
        named = dict(model.named_parameters())
        return loss.item(), logits.detach().float().cpu(), {k: named[k].grad.double().cpu() for k in g64}

    # ---- f32: the parity mode
    loss, logits, grads = run(False)
    assert abs(loss - loss64) < 1e-4 * abs(loss64), (loss, loss64)
    rel = ((logits.double() - logits64).abs().max() / logits64.abs().max()).item()
    assert rel < 1e-4, rel
    dev_gpu = {k: abs(grads[k].norm().item() / g64[k].norm().item() - 1) for k in g64}
    dev_f32 = {k: max(abs(g32[k].norm().item() / g64[k].norm().item() - 1) for g32 in f32_runs) for k in g64}
    print(name, 'f32 |g| deviation: hip', {k: '%.1e' % v for k, v in dev_gpu.items()})
    print(name, 'f32 |g| deviation: cpu f32 oracle, worst of 3 summation orders', {k: '%.1e' % v for k, v in dev_f32.items()})
    bar = max(3e-4, 2 * max(dev_f32.values()))
    assert max(dev_gpu.values()) <= bar, (dev_gpu, dev_f32)
    def miss(a, ref):                           # 1 - cosine, whole tensors
        return 1.0 - ((a * ref).sum() / (a.norm() * ref.norm())).item()
    miss_f32 = max(miss(g32[k], g64[k]) for g32 in f32_runs for k in g64)
    for k in g64:                               # directions: same calibration
        assert miss(grads[k], g64[k]) <= max(1e-6, 4 * miss_f32), (k, miss(grads[k], g64[k]), miss_f32)

    # ---- bf16 autocast: the bench dtype.  With 83 k rows behind every BatchNorm statistic what is left
    # is bf16 rounding (2^-9 per stored activation / gradient element) AMPLIFIED by the conditioning of
    # this randomly initialised 49-layer net: the f32 runs above (rounding 2^-24) already miss the f64
    # gradients by up to 4.5e-4 on the SAME parameters the bf16 run misses most (a BatchNorm gamma whose
    # gradient is a cancelling sum over 37 k rows: amplification ~10^4), HIP and CPU oracle alike.
    # Measured at this size: cosine 0.908-0.9999, |g| within 11 % (MinkUNet is the worse of the two).  To tell rounding from a bias of the
    # statistics path, the step is ALSO run with the convolution-epilogue tile statistics switched off
    # (BatchNorm then makes its own f64 pass over the stored matrix): both runs must sit at the same
    # distance from the f64 gradients.
    loss16, logits16, grads16 = run(True)
    assert abs(loss16 - loss64) < 1e-2 * abs(loss64), (loss16, loss64)

    def report_of(gr):
        rep = {}
        for k in g64:
            cos = ((gr[k] * g64[k]).sum() / (gr[k].norm() * g64[k].norm())).item()
            rep[k] = (round(cos, 5), round(gr[k].norm().item() / g64[k].norm().item(), 4))
        return rep
    report = report_of(grads16)
    print(name, 'bf16 (cosine, |g| ratio):', report)
    no_stats[0] = True
    _, _, grads16_own = run(True)
    no_stats[0] = False
    report_own = report_of(grads16_own)
    print(name, 'bf16, BatchNorm statistics by their own pass:', report_own)
    for k, (cos, ratio) in report.items():
        if k in ('classifier.0.weight', 'up4.1.1.net.3.kernel'):      # the last layers: little depth to amplify
            assert cos >= 0.999 and abs(ratio - 1) <= 0.01, report
        # (|g| of one BatchNorm gamma -- a cancelling sum over 37 k rows -- has come out between 0.89 and 1.15 of the f64
        # value from one association of the f32 sums to the next: round 4's split of the coarse levels' offsets moved it
        # from 0.89 to 1.14 while every operation of the same step sits within half a bf16 ulp of the oracle's operator,
        # tests/test_teacher_forced_gpu.py.  These are sanity bounds on a chaotic quantity, not the parity statement.)
        assert cos >= 0.87 and abs(ratio - 1) <= 0.2, report
        cos_own, ratio_own = report_own[k]
        assert abs(cos - cos_own) <= 0.04 and abs(ratio - ratio_own) <= 0.15, (k, report[k], report_own[k])


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_bf16_train_step_matches_the_bf16_emulating_oracle(name, one_scan):
    """The benchmarked dtype against a reference AT ITS OWN PRECISION (round 4): the oracle model with every
    activation and activation gradient rounded to bf16 exactly where the HIP path stores bf16 (oracle/models_ref.py
    emulate_bf16: conv / Linear / voxel-exchange outputs, BatchNorm outputs, the fused relu(bn + shortcut) and
    point-branch sums rounded once, each consumer's gradient rounded before autograd sums them, bf16 weight operands,
    f32 weight gradients).  One ~120 k-point scan (~83 k voxels), whole train step.

    Forward: loss within 1e-3, every logit within 4 bf16 ulps of the largest logit.

    Gradients -- the FINDING of this test.  Two runs of the emulating oracle ITSELF that differ only in the arithmetic
    BETWEEN the storage points (float64 against float32 accumulation, or float32 with the rows permuted) already
    disagree on the deep parameters at cosine 0.93-0.99 (|g| within 2-9 %), hardly closer than either is to the plain
    f64 run; permuting the rows under float64 changes nothing (cosine 1.00000).  With bf16 storage a last-bit
    difference of an accumulation flips the rounding of a few stored elements by a whole bf16 ulp, and the randomly
    initialised 49-layer train-mode-BatchNorm net amplifies that like any other perturbation: no implementation whose
    f32 sums run in another order than its reference's can reach cosine 0.999 end to end, the reference against
    itself included.  So the end-to-end gradient bar is calibrated on the reference's own spread (two samples of a
    chaotic quantity: a factor of four, floors 1e-4 / 6 %) -- a sanity bound, not the parity statement; the last
    layers (nothing upstream to amplify) are held to cosine 0.999 / 1 %.  The parity statement for the bf16 kernels
    inside the whole step is tests/test_teacher_forced_gpu.py: every stored tensor of the SAME step against the
    emulating reference operator applied to the step's own stored inputs, where nothing amplifies.  The distance to the
    plain f64 run is printed as information."""
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.train_step import forward_backward
    from oracle import tsref
    from oracle.models_ref import MinkUNetRef, SPVCNNRef, emulate_bf16
    from weights import fill_state_dict
    coords, feats, labels = one_scan
    torch.set_num_threads(min(32, max(torch.get_num_threads(), (__import__('os').cpu_count() or 8))))
    ref_cls = {'spvcnn': SPVCNNRef, 'minkunet': MinkUNetRef}[name]
    keys = GKEYS + (['point_transforms.1.0.weight'] if name == 'spvcnn' else [])

    def oracle(dt, emulate, perm=None):
        model = fill_state_dict(ref_cls(19)).to(dt).train()
        if hasattr(model, 'dropout'):
            model.dropout.p = 0.0
        if emulate:
            emulate_bf16(model)
        c, f, lab = (coords, feats, labels) if perm is None else (coords[perm], feats[perm], labels[perm])
        logits, _ = model(tsref.SparseTensor(f.clone().to(dt), c.clone()))
        loss = torch.nn.functional.cross_entropy(logits, lab, ignore_index=255, reduction='mean')
        loss.backward()
        named = dict(model.named_parameters())
        return loss.item(), logits.detach().double(), {k: named[k].grad.detach().double() for k in keys if k in named}

    loss_e, logits_e, g_e = oracle(torch.float64, True)
    assert torch.equal(logits_e, logits_e.to(torch.bfloat16).double())          # the emulation really stores bf16
    perm = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(1))
    spread = [oracle(torch.float32, True)[2], oracle(torch.float32, True, perm)[2]]
    g_64 = oracle(torch.float64, False)[2]
    model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).to(DEV).train()
    if hasattr(model, 'dropout'):
        model.dropout.p = 0.0
    loss, logits = forward_backward(model, feats.to(DEV), coords.to(DEV), labels.to(DEV), autocast=True)
    named = dict(model.named_parameters())
    grads = {k: named[k].grad.double().cpu() for k in g_e}
    logits = logits.detach().double().cpu()

    def dist(a, ref):                           # (1 - cosine, | |a| / |ref| - 1 |), whole tensors
        return (1.0 - ((a * ref).sum() / (a.norm() * ref.norm())).item(), abs(a.norm().item() / ref.norm().item() - 1.0))
    ulp = float(logits_e.abs().max()) * 2.0 ** -8           # one bf16 ulp at the largest logit
    d_emul = float((logits - logits_e).abs().max())
    print(name, 'bf16 HIP vs bf16-emulating oracle: loss %.6f / %.6f, max |dlogit| %.2f ulp' % (loss.item(), loss_e, d_emul / ulp))
    assert abs(loss.item() - loss_e) < 1e-3 * abs(loss_e), (loss.item(), loss_e)
    assert d_emul <= 4 * ulp, d_emul / ulp
    print(name, '  %-30s %-22s %-22s %s' % ('parameter: (1-cos, |g| dev)', 'HIP ~ emulation(f64)', 'emulation(f32) ~ (f64)', 'HIP ~ plain f64'))
    for k in g_e:
        hip = dist(grads[k], g_e[k])
        ref = (max(dist(g[k], g_e[k])[0] for g in spread), max(dist(g[k], g_e[k])[1] for g in spread))
        info = dist(grads[k], g_64[k])
        print(name, '  %-30s %.1e %.1e        %.1e %.1e        %.1e %.1e' % (k, hip[0], hip[1], ref[0], ref[1], info[0], info[1]))
        if k in ('classifier.0.weight', 'up4.1.1.net.3.kernel'):        # the last layers: little depth to amplify
            assert hip[0] <= 1e-3 and hip[1] <= 0.01, (k, hip)
        assert hip[0] <= 4 * ref[0] + 1e-4 and hip[1] <= 4 * ref[1] + 6e-2, (k, hip, ref)


def _f64_wgrad(a, b, pairs, koff, a_col, k):
    ko = koff.cpu().tolist()
    ref = torch.zeros(k, a.shape[1], b.shape[1], dtype=torch.float64, device=a.device)
    for kk in range(k):
        if ko[kk + 1] == ko[kk]:
            continue
        if pairs is None:
            ia = ib = torch.arange(ko[kk], ko[kk + 1], device=a.device)
        else:
            pr = pairs[ko[kk]:ko[kk + 1]].long()
            ia, ib = (pr[:, 1], pr[:, 0]) if a_col else (pr[:, 0], pr[:, 1])
        ref[kk] = a[ia].double().t() @ b[ib].double()
    return ref


@pytest.mark.parametrize('ca,cb,k', [(96, 96, 27), (128, 96, 27), (128, 96, 1), (32, 32, 27)])
def test_wgrad_plan_of_the_bench_batch(ca, cb, k, bench_batch):
    """lidal_conv_wgrad with the plan bench.py's level-0 layers get (396 662 rows) vs f64."""
    from lidal_amd import backend as B
    from lidal_amd.nn import functional as F
    from lidal_amd.nn.functional.conv import wgrad_scratch
    coords = bench_batch
    n = coords.shape[0]
    assert n > 350000
    g = torch.Generator().manual_seed(ca + cb + k)
    a = torch.randn(n, ca, generator=g).to(DEV).bfloat16()
    b = torch.randn(n, cb, generator=g).to(DEV).bfloat16()
    if k == 27:
        km, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
        pairs, koff = km._nbmaps_cap, km.koff
    else:
        pairs, koff = None, torch.tensor([0, n], dtype=torch.int64, device=DEV)

    def run():
        gw = torch.full((k, ca, cb), float('nan'), dtype=torch.float32, device=DEV)
        part = wgrad_scratch(n, n, k, ca, cb, a.dtype, DEV)
        B.check(B.lib().lidal_conv_wgrad(B.ptr(a), B.ptr(b), n, n, B.ptr(pairs), B.ptr(koff), 0, B.ptr(gw),
                                         B.ptr(part), part.shape[0], k, ca, cb, B.dtype_code(a.dtype),
                                         B.stream()), 'conv_wgrad')
        return gw
    got, again = run(), run()
    assert torch.equal(got, again)
    ref = _f64_wgrad(a, b, pairs, koff, 0, k)
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err


@pytest.mark.parametrize('ci,co', [(96, 96), (128, 96), (32, 32)])
def test_conv_forward_and_flipped_dgrad_of_the_bench_batch(ci, co, bench_batch):
    """Forward and data gradient (the `kflip` walk over nbr_out of a symmetric map) of a level-0
    k3 layer at ~4e5 rows, bf16 operands, against f64 index_select / matmul / index_add on the GPU."""
    import lidal_amd
    from lidal_amd.nn import functional as F
    coords = bench_batch
    n = coords.shape[0]
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, generator=g).to(DEV).bfloat16().requires_grad_(True)
    w = (torch.randn(27, ci, co, generator=g) * 0.05).to(DEV).bfloat16().requires_grad_(True)
    gy = torch.randn(n, co, generator=g).to(DEV).bfloat16()
    st = lidal_amd.SparseTensor(x, coords, 1)
    y = F.conv3d(st, w, 3).F
    y.backward(gy)
    km = st.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    assert km.symmetric
    maps, ko = km.nbmaps.long(), km.koff.cpu().tolist()
    xd, wd, gd = x.detach().double(), w.detach().double(), gy.double()
    y_ref = torch.zeros(n, co, dtype=torch.float64, device=DEV)
    gx_ref = torch.zeros(n, ci, dtype=torch.float64, device=DEV)
    gw_ref = torch.zeros(27, ci, co, dtype=torch.float64, device=DEV)
    for kk in range(27):
        pr = maps[ko[kk]:ko[kk + 1]]
        if pr.shape[0] == 0:
            continue
        y_ref.index_add_(0, pr[:, 1], xd[pr[:, 0]] @ wd[kk])
        gx_ref.index_add_(0, pr[:, 0], gd[pr[:, 1]] @ wd[kk].t())
        gw_ref[kk] = xd[pr[:, 0]].t() @ gd[pr[:, 1]]

    def rel(a, ref):
        return ((a.double() - ref).abs().max() / ref.abs().max()).item()
    # outputs are ROUNDED to bf16 (2^-9 relative per element) from f32 accumulators: elementwise bound
    def close(a, ref):
        return bool(((a.double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 1e-4 * ref.abs().max()).all())
    assert close(y, y_ref), rel(y, y_ref)
    assert close(x.grad, gx_ref), rel(x.grad, gx_ref)
    assert close(w.grad, gw_ref), rel(w.grad, gw_ref)       # returned in the weight's dtype (bf16 here)
    # f32 parameters under autocast (what the model does): the weight gradient comes back unrounded
    xs = x.detach().float().requires_grad_(True)
    ws = w.detach().float().requires_grad_(True)
    st = lidal_amd.SparseTensor(xs, coords, 1)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y32 = F.conv3d(st, ws, 3).F
    y32.backward(gy)
    assert ws.grad.dtype == torch.float32
    assert rel(ws.grad, gw_ref) < 1e-4, rel(ws.grad, gw_ref)


def _frames_120k(n_frames):
    from lidal_amd import synth
    frames = synth.make_sequence(n_frames, n_points=120000, seed=7122)
    rng = np.random.default_rng(5)
    probs = []
    for f in frames:
        w = f['world']
        lg = rng.standard_normal((w.shape[0], 19)).astype(np.float32) + np.sin(w[:, :1] * 0.7).astype(np.float32) * 2
        p = np.exp(lg - lg.max(1, keepdims=True))
        probs.append((p / p.sum(1, keepdims=True)).astype(np.float32))
    return frames, probs


@pytest.mark.parametrize('nei,n_frames,queries', [(10, 13, (0, 6, 12)), (24, 27, (0, 13, 26))])
def test_scoring_at_120k_points_matches_oracle(nei, n_frames, queries):
    from lidal_amd.score import FrameBank, interframe
    from oracle import scoring_ref
    frames, probs = _frames_120k(n_frames)
    worlds = [f['world'] for f in frames]
    assert worlds[0].shape[0] >= 100000
    bank = FrameBank(0.1)
    for p, w in zip(probs, worlds):
        bank.add(torch.from_numpy(w).to(DEV), torch.from_numpy(p).to(DEV))
    matched = 0
    for i in queries:
        sv2point = frames[i]['sv2point']
        rd, re, rn, rc, pd, pe = scoring_ref.score_frame(i, probs, worlds, sv2point, nei, 0.1, return_points=True)
        interd, intere, cnt = interframe.score_points(bank, i, nei)
        matched += int((cnt > 0).sum())
        assert np.abs(interd.cpu().numpy() - pd).max() <= 1e-4 * max(np.abs(pd).max(), 1e-12)
        assert np.abs(intere.cpu().numpy() - pe).max() <= 1e-4 * np.abs(pe).max()
        ptr, idx, lens = interframe.sv_csr(sv2point, DEV)
        d, e, c = interframe.score_frame(bank, i, ptr, idx, nei)
        assert np.allclose(d.cpu().numpy(), rd, rtol=1e-4, atol=1e-7)
        assert np.allclose(e.cpu().numpy(), re, rtol=1e-4, atol=1e-7)
        assert np.allclose(c.cpu().numpy(), rc, rtol=1e-5, atol=1e-5)
        assert np.array_equal(lens, rn)
    assert matched > 50000, matched


def test_bf16_inference_against_f32_inference_on_scores_and_selection(capsys):
    """What bf16 conv operands at inference do to the OUTPUT of the scoring pipeline (the reference infers in fp32,
    score/prob_inference.py:91-113; bench.py's `secondary.value` is the f32 figure, the bf16 one rides beside it):
    13 frames of ~120 k points, 8 views each, SPVCNN, window 10.  Printed and bounded: the largest deviation of
    sv_interds / sv_interes relative to the largest score, Spearman's rank correlation of the divergences (the selection
    is rank-based, LiDAL.py:233-235) and the fraction of supervoxels to which select() gives the same flag."""
    from scipy.stats import spearmanr
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN
    from lidal_amd.score import collect_sequence, interframe, score_sequence
    from lidal_amd.score.selection import select
    n_frames, nei = 13, 10
    frames = synth.make_sequence(n_frames, n_points=120000, seed=7122)
    dev_frames = []
    for i, f in enumerate(frames):
        sb = synth.make_score_batch(f['points'], f['intensity'], np.random.default_rng([7122, 99, 0, i]), inf_reps=8)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], DEV)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(DEV), 'feats': torch.from_numpy(sb['feats_v_b']).to(DEV),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(DEV),
                           'world': torch.from_numpy(f['world']).to(DEV), 'sv_ptr': ptr, 'sv_idx': idx})
    torch.manual_seed(7122)
    model = SPVCNN(19).to(DEV).eval()
    got = {}
    for name, autocast in (('f32', False), ('bf16', True)):
        scores = score_sequence(model, dev_frames, 0, n_frames, nei_num=nei, dis_thresh=0.1, inf_reps=8, autocast=autocast)
        rows = collect_sequence(scores, [f['sv_id'] for f in frames], [d['sv_ptr'] for d in dev_frames], 0, n_frames)
        got[name] = {k: np.concatenate([r[j] for r in rows]) for j, k in enumerate(('id', 'd', 'e', 'n', 'c'))}
    a, b = got['f32'], got['bf16']
    assert np.array_equal(a['id'], b['id']) and np.array_equal(a['n'], b['n']) and np.array_equal(a['c'], b['c'])
    dev_d = np.abs(a['d'] - b['d']).max() / np.abs(a['d']).max()
    dev_e = np.abs(a['e'] - b['e']).max() / np.abs(a['e']).max()
    rho_d = spearmanr(a['d'], b['d'])[0]
    rho_e = spearmanr(a['e'], b['e'])[0]
    n_sv = a['id'].shape[0]
    flags0 = np.zeros(n_sv, dtype=int)
    budget = int(a['n'].sum())                  # 1 % of it per pass: a few supervoxels each
    fa = select(flags0, a['d'], a['e'], a['n'], a['c'], 30 * budget)
    fb = select(flags0, b['d'], b['e'], b['n'], b['c'], 30 * budget)
    agree = float((fa == fb).mean())
    with capsys.disabled():
        print('\n[bf16 vs f32 inference, %d supervoxels] max |d sv_interds| / max = %.3e, max |d sv_interes| / max = %.3e, '
              'Spearman rho (divergence) = %.5f, (entropy) = %.5f, select() flags equal on %.1f %% (%d AL / %d SL picks under f32)'
              % (n_sv, dev_d, dev_e, rho_d, rho_e, 100 * agree, int((fa == 1).sum()), int((fa == 2).sum())))
    assert (fa == 1).sum() >= 3 and (fa == 2).sum() >= 3
    # measured on MI355X (randomly initialised SPVCNN: near-uniform probabilities, so the entropies hardly move; a
    # trained network's sharper probabilities would move more): 8.6e-4 / 8.1e-8, rho 0.99999 / 0.99997, 99.6 % of flags
    assert dev_e < 1e-4 and rho_e > 0.999, (dev_e, rho_e)
    assert dev_d < 1e-2 and rho_d > 0.999, (dev_d, rho_d)
    assert agree >= 0.97, agree
