"""Scoring parity on the GPU: the fused view-mean softmax and the inter-frame divergence /
entropy scorer against (a) golden outputs of the REFERENCE's own worker_func
(tests/golden/scoring_small.npz) and (b) the CPU oracle on fresh seeded frames."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bank(probs, worlds, dis):
    from lidal_amd.score import FrameBank
    bank = FrameBank(dis)
    for p, w in zip(probs, worlds):
        bank.add(torch.from_numpy(np.ascontiguousarray(w)).to(DEV),
                 torch.from_numpy(np.ascontiguousarray(p)).to(DEV))
    return bank


def test_interframe_matches_reference_worker_func(golden_dir):
    from lidal_amd.score import interframe
    g = np.load(os.path.join(golden_dir, 'scoring_small.npz'))
    nei, dis = int(g['nei_num']), float(g['dis_thresh'])
    bank = _bank(g['probs'], g['worlds'], dis)
    n_frames = len(bank)
    interd, intere, cnt = interframe.score_points(bank, 0, nei)
    ref_d, ref_e = g['interd_points_f0'], g['intere_points_f0']
    assert np.abs(interd.cpu().numpy() - ref_d).max() <= 1e-4 * max(np.abs(ref_d).max(), 1e-12)
    assert np.abs(intere.cpu().numpy() - ref_e).max() <= 1e-4 * np.abs(ref_e).max()
    assert int((cnt > 0).sum()) > 50          # the fixture really exercises matches
    for i in range(n_frames):
        ptr, idx, lens = interframe.sv_csr(list(g['sv2point'][i]), DEV)
        d, e, c = interframe.score_frame(bank, i, ptr, idx, nei)
        assert np.allclose(d.cpu().numpy(), g['sv_interds'][i], rtol=1e-4, atol=1e-7), i
        assert np.allclose(e.cpu().numpy(), g['sv_interes'][i], rtol=1e-4, atol=1e-7), i
        assert np.allclose(c.cpu().numpy(), g['sv_centers'][i], rtol=1e-5, atol=1e-5), i
        assert np.array_equal(lens, g['sv_pnums'][i])


def test_interframe_matches_oracle_on_ragged_frames():
    """Frames of different sizes, 10-neighbour window (BASELINE.json config 5)."""
    from lidal_amd import synth
    from lidal_amd.score import interframe
    from oracle import scoring_ref
    frames = synth.make_sequence(13, n_points=None, seed=3, step=0.8, n_beams=24, n_az=256)
    rng = np.random.default_rng(0)
    probs, worlds = [], []
    for f in frames:
        keep = rng.random(f['world'].shape[0]) < rng.uniform(0.6, 1.0)
        w = f['world'][keep]
        lg = rng.standard_normal((w.shape[0], 19)) + np.sin(w[:, :1] * 0.7) * 2
        p = np.exp(lg - lg.max(1, keepdims=True))
        probs.append((p / p.sum(1, keepdims=True)).astype(np.float32))
        worlds.append(w)
    bank = _bank(probs, worlds, 0.1)
    matched = 0
    for i in (0, 6, 12):
        sv2point = synth.angular_supervoxels(worlds[i].astype(np.float32), 20)
        rd, re, rn, rc, pd, pe = scoring_ref.score_frame(i, probs, worlds, sv2point, 10, 0.1,
                                                         return_points=True)
        interd, intere, cnt = interframe.score_points(bank, i, 10)
        matched += int((cnt > 0).sum())
        assert np.abs(interd.cpu().numpy() - pd).max() <= 1e-4 * max(np.abs(pd).max(), 1e-12)
        assert np.abs(intere.cpu().numpy() - pe).max() <= 1e-4 * np.abs(pe).max()
        ptr, idx, _ = interframe.sv_csr(sv2point, DEV)
        d, e, c = interframe.score_frame(bank, i, ptr, idx, 10)
        assert np.allclose(d.cpu().numpy(), rd, rtol=1e-4, atol=1e-7)
        assert np.allclose(e.cpu().numpy(), re, rtol=1e-4, atol=1e-7)
    assert matched > 1000


def test_view_mean_softmax_matches_oracle():
    from lidal_amd.score.prob_inference import view_mean_softmax
    from oracle.harness_ref import inference_post
    g = torch.Generator().manual_seed(0)
    reps, p, c, nv = 8, 5000, 19, 3000
    logits = torch.randn(reps * nv, c, generator=g) * 3
    inverse = torch.cat([torch.randint(0, nv, (p,), generator=g) + v * nv for v in range(reps)])
    prob_ref, pred_ref = inference_post(logits, inverse, reps)
    prob, pred = view_mean_softmax(logits.to(DEV), inverse.to(DEV), reps)
    assert np.abs(prob.cpu().numpy() - prob_ref).max() <= 1e-4 * np.abs(prob_ref).max()
    assert (pred.cpu().numpy() == pred_ref).mean() > 0.999


def test_prob_inference_end_to_end_vs_oracle():
    """8 augmented views of one small frame through MinkUNet + fused post-processing vs the oracle
    model + score/prob_inference.py:100-113 restatement."""
    import lidal_amd
    from lidal_amd import synth
    from lidal_amd.network import MinkUNet
    from lidal_amd.score import infer_frame
    from oracle import harness_ref
    from oracle.models_ref import MinkUNetRef
    from weights import fill_state_dict
    rng = np.random.default_rng(1)
    world = synth.make_world(5)
    pts, inten = synth.raycast_scan(world, (20.0, 0.0), rng, n_beams=16, n_az=128)
    batch = synth.make_score_batch(pts, inten, rng, inf_reps=8)
    coords = torch.from_numpy(batch['coords_v_b'])
    feats = torch.from_numpy(batch['feats_v_b'])
    inverse = torch.from_numpy(batch['inverse_indices_b'])
    model = fill_state_dict(MinkUNet(19)).eval()
    sd = model.state_dict()
    prob, pred = infer_frame(model.to(DEV), coords.to(DEV), feats.to(DEV), inverse.to(DEV), 8)
    # oracle: the same architecture restated on the CPU operators (oracle/models_ref.py)
    ref_model = MinkUNetRef(19)
    ref_model.load_state_dict(sd, strict=True)
    ref_model.eval()
    with torch.no_grad():
        logits_ref, _ = harness_ref.forward(ref_model, feats, coords)
    prob_ref, pred_ref = harness_ref.inference_post(logits_ref, inverse, 8)
    assert prob.shape == prob_ref.shape
    assert np.abs(prob.cpu().numpy() - prob_ref).max() <= 1e-4 * np.abs(prob_ref).max() + 1e-6
    assert (pred.cpu().numpy() == pred_ref).mean() > 0.995


def test_confusion_matrix_matches_reference_bincount():
    """evaluate.py:100-109 + utils/iou_sk.py:14-19 restated in numpy vs the device kernel."""
    from lidal_amd.evaluate import confusion_accumulate, iou_from_confusion
    g = torch.Generator().manual_seed(4)
    nv, p, c = 5000, 40000, 19
    logits = torch.randn(nv, c, generator=g)
    inverse = torch.randint(0, nv, (p,), generator=g)
    labels = torch.randint(0, c, (p,), generator=g)
    labels[torch.rand(p, generator=g) < 0.1] = 255
    pred = logits[inverse].max(1)[1].numpy()
    gt = labels.numpy()
    idx = gt < 100
    ref = np.bincount(pred[idx] * 19 + gt[idx], minlength=361).reshape(19, 19).astype(np.int32)
    conf = torch.zeros((c, c), dtype=torch.int32, device=DEV)
    confusion_accumulate(conf, logits.to(DEV), inverse.to(DEV), labels.to(DEV))
    confusion_accumulate(conf, logits.to(DEV), inverse.to(DEV), labels.to(DEV))       # accumulates
    assert np.array_equal(conf.cpu().numpy(), 2 * ref)
    ious, miou = iou_from_confusion(ref)
    assert 0.0 < miou < 0.2


def test_score_sequence_pipeline_matches_oracle_end_to_end():
    """prob_inference + LiDAL scoring in one call (lidal_amd.score.score_sequence) vs the oracle:
    MinkUNetRef forward on CPU, score/prob_inference.py:100-113 restated, then scoring_ref."""
    from lidal_amd import synth
    from lidal_amd.network import MinkUNet
    from lidal_amd.score import interframe, score_sequence
    from oracle import harness_ref, scoring_ref
    from oracle.models_ref import MinkUNetRef
    from weights import fill_state_dict
    n_frames, nei = 5, 4
    frames = synth.make_sequence(n_frames, n_points=None, seed=21, step=0.5, n_beams=12, n_az=96)
    rng = np.random.default_rng(3)
    model = fill_state_dict(MinkUNet(19)).eval()
    ref_model = MinkUNetRef(19)
    ref_model.load_state_dict(model.state_dict(), strict=True)
    ref_model.eval()
    model = model.to(DEV)
    dev_frames, probs_ref = [], []
    for f in frames:
        sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=2)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], DEV)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(DEV),
                           'feats': torch.from_numpy(sb['feats_v_b']).to(DEV),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(DEV),
                           'world': torch.from_numpy(f['world']).to(DEV), 'sv_ptr': ptr, 'sv_idx': idx})
        with torch.no_grad():
            lo, _ = harness_ref.forward(ref_model, torch.from_numpy(sb['feats_v_b']),
                                        torch.from_numpy(sb['coords_v_b']))
        probs_ref.append(harness_ref.inference_post(lo, torch.from_numpy(sb['inverse_indices_b']), 2)[0])
    out = score_sequence(model, dev_frames, 0, n_frames, nei_num=nei, dis_thresh=0.1, inf_reps=2)
    # scoring beside the inference of the following frames (on the table builder's stream, on a third stream) and scoring
    # after it: the same numbers bit for bit
    from lidal_amd.score import pipeline
    saved = pipeline._OVERLAP, pipeline._STREAM
    try:
        runs = []
        for mode, stream in (('1', 'tables'), ('1', 'third'), ('0', 'auto')):
            pipeline._OVERLAP, pipeline._STREAM = mode, stream
            runs.append(score_sequence(model, dev_frames, 0, n_frames, nei_num=nei, dis_thresh=0.1, inf_reps=2))
    finally:
        pipeline._OVERLAP, pipeline._STREAM = saved
    torch.cuda.synchronize()
    for r in runs:
        for a, o in zip(r, out):
            for u, w in zip(a, o):
                assert torch.equal(u, w)
    worlds = [f['world'] for f in frames]
    for i in range(n_frames):
        rd, re, rn, rc = scoring_ref.score_frame(i, probs_ref, worlds, frames[i]['sv2point'], nei, 0.1)
        d, e, c = out[i]
        assert np.allclose(d.cpu().numpy(), rd, rtol=2e-3, atol=1e-6), i     # probabilities differ by
        assert np.allclose(e.cpu().numpy(), re, rtol=2e-3, atol=1e-6), i     # 1e-4 before the KL
        assert np.allclose(c.cpu().numpy(), rc, rtol=1e-5, atol=1e-5), i


def test_infer_frame_returns_the_point_features_on_request():
    """prob_inference.py:103-105,116-118 (`outfeat`, r_id == 0 / ReDAL / CSET): the [P, 96] view-mean of
    feat[inverse_indices], against the oracle model + the reference expression."""
    from lidal_amd import synth
    from lidal_amd.network import MinkUNet
    from lidal_amd.score import infer_frame
    from oracle import harness_ref
    from oracle.models_ref import MinkUNetRef
    from weights import fill_state_dict
    rng = np.random.default_rng(2)
    world = synth.make_world(6)
    pts, inten = synth.raycast_scan(world, (20.0, 0.0), rng, n_beams=16, n_az=128)
    batch = synth.make_score_batch(pts, inten, rng, inf_reps=8)
    coords, feats = torch.from_numpy(batch['coords_v_b']), torch.from_numpy(batch['feats_v_b'])
    inverse = torch.from_numpy(batch['inverse_indices_b'])
    model = fill_state_dict(MinkUNet(19)).eval()
    ref_model = MinkUNetRef(19)
    ref_model.load_state_dict(model.state_dict(), strict=True)
    ref_model.eval()
    prob, pred, feat = infer_frame(model.to(DEV), coords.to(DEV), feats.to(DEV), inverse.to(DEV), 8, return_feat=True)
    prob2, pred2 = infer_frame(model, coords.to(DEV), feats.to(DEV), inverse.to(DEV), 8)
    assert torch.equal(prob, prob2) and torch.equal(pred, pred2)
    with torch.no_grad():
        _, feat_ref = ref_model(__import__('oracle').tsref.SparseTensor(feats, coords))
    want = feat_ref[inverse].numpy().reshape(8, -1, feat_ref.shape[1]).mean(0)
    assert feat.shape == want.shape
    assert np.abs(feat.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()


def test_256_frame_sequence_scores_match_the_oracle_at_both_ends_and_in_the_middle():
    """BASELINE.json configs[3] at its stated length: ONE sequence of 256 frames through score_sequence on one GPU (the
    FrameBank holding every frame, scoring queued beside the inference of the frames that follow), then frames 0, 127 and
    255 against oracle.scoring_ref -- the first and the last one read the wrap-rule frames of
    /root/reference/score/sv_level/LiDAL.py:41-42 (frame 0: 6..10 stand in for -1..-5; frame 255: 249..245 for 256..260).
    The oracle scores them from the probabilities of their own windows, inferred frame by frame with infer_frame (bitwise
    what the pipeline fed its bank: same kernels, same inputs)."""
    from lidal_amd import synth
    from lidal_amd.network import MinkUNet
    from lidal_amd.score import infer_frame, interframe, score_sequence
    from oracle import scoring_ref
    from weights import fill_state_dict
    n_frames, nei, reps = 256, 10, 2
    frames = synth.make_sequence(n_frames, n_points=None, seed=33, step=0.3, n_beams=16, n_az=192)
    rng = np.random.default_rng(5)
    model = fill_state_dict(MinkUNet(19)).eval().to(DEV)
    dev_frames = []
    for f in frames:
        sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=reps)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], DEV)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(DEV),
                           'feats': torch.from_numpy(sb['feats_v_b']).to(DEV),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(DEV),
                           'world': torch.from_numpy(f['world']).to(DEV), 'sv_ptr': ptr, 'sv_idx': idx})
    out = score_sequence(model, dev_frames, 0, n_frames, nei_num=nei, dis_thresh=0.1, inf_reps=reps)
    torch.cuda.synchronize()
    assert len(out) == n_frames
    assert interframe.neighbour_ids(0, n_frames, nei) == [6, 7, 8, 9, 10, 1, 2, 3, 4, 5]
    assert interframe.neighbour_ids(255, n_frames, nei) == [254, 253, 252, 251, 250, 249, 248, 247, 246, 245]
    matched = 0
    for i in (0, 127, 255):
        window = scoring_ref.neighbour_ids(i, n_frames, nei) + [i]
        assert window[:-1] == interframe.neighbour_ids(i, n_frames, nei)
        probs, worlds = [None] * n_frames, [None] * n_frames
        for j in window:
            d = dev_frames[j]
            probs[j] = infer_frame(model, d['coords'], d['feats'], d['inverse'], reps)[0].cpu().numpy()
            worlds[j] = frames[j]['world']
        rd, re, rn, rc, pd, pe = scoring_ref.score_frame(i, probs, worlds, frames[i]['sv2point'], nei, 0.1,
                                                         return_points=True)
        matched += int((pd > 0).sum())
        d, e, c = out[i]
        assert np.allclose(d.cpu().numpy(), rd, rtol=1e-4, atol=1e-7), i
        assert np.allclose(e.cpu().numpy(), re, rtol=1e-4, atol=1e-7), i
        assert np.allclose(c.cpu().numpy(), rc, rtol=1e-5, atol=1e-5), i
    assert matched > 100, matched          # the windows really overlap
    # the sequence in two halves of a sharded run would read the same frames: the halo plan at 8 blocks of 32 covers
    # every frame's window (score/sharding.py: needed_frames)
    from lidal_amd.score.sharding import frame_range, needed_frames
    for r in range(8):
        need = set(needed_frames(n_frames, 8, r, nei))
        for i in frame_range(n_frames, 8, r):
            assert set(interframe.neighbour_ids(i, n_frames, nei)) | {i} <= need, (r, i)


def test_cell_ordered_queries_give_the_same_scores_bit_for_bit():
    """lidal_interframe_score_ordered (round 6; off by default: no gain): the queries of a frame taken in the cell order of
    its own grid -- the same per-point arithmetic, so every output equals the scan-order call's."""
    from lidal_amd import synth
    from lidal_amd.score import interframe
    frames = synth.make_sequence(7, n_points=None, seed=8, step=0.6, n_beams=24, n_az=256)
    rng = np.random.default_rng(1)
    probs, worlds = [], []
    for f in frames:
        w = f['world']
        lg = rng.standard_normal((w.shape[0], 19)) + np.cos(w[:, 1:2] * 0.9) * 2
        p = np.exp(lg - lg.max(1, keepdims=True))
        probs.append((p / p.sum(1, keepdims=True)).astype(np.float32))
        worlds.append(w)
    bank = _bank(probs, worlds, 0.1)
    saved = interframe.CELL_ORDER
    try:
        outs = []
        for order in (False, True):
            interframe.CELL_ORDER = order
            outs.append([interframe.score_points(bank, i, 4) for i in (0, 3, 6)])
    finally:
        interframe.CELL_ORDER = saved
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert int((outs[0][1][2] > 0).sum()) > 100
