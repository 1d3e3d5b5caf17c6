"""One scoring pass over a sequence, frame-sharded: the GPU counterpart of running
/root/reference/score/prob_inference.py (per-frame 8-view inference, :91-133) followed by
score/sv_level/LiDAL.py:185-218 (per-frame inter-frame divergence / entropy per supervoxel) and
the hand-off to the single-process selection (:208-222 scatter into the global arrays, :225-330).

Each rank owns a contiguous block of frames (dataset/sk_dataloader.py:196-198); probabilities and
world coordinates are exchanged by one all-gather each (replacing the .npy / KD-tree pickle
hand-off), after which every rank scores its own frames against full +-nei windows.  The
per-supervoxel results (20 x 7 numbers per frame) then travel to rank 0 in one more collective;
selection is host Python on rank 0 (replicas only).
"""
import numpy as np
import torch
import torch.distributed as dist

from .interframe import FrameBank, score_frame
from .prob_inference import infer_frame
from .selection import select
from .sharding import HaloExchange, gather_frames, is_sharded

__all__ = ['score_sequence', 'collect_sequence', 'ScoreBoard']

import os as _os
# Scoring beside the inference of the frames that follow: on WHICH stream decides whether it pays.  Measured on bound
# processes, round 3's tree beside it, frames/s at nei 10 / 24 (scripts/gpu/archive/sec_ab.sh, sec_ab2.sh; two boxes, 2-3
# repetitions): round 3 125-129 / 117-120; planned inference, scoring AFTER it 131-133 / 126; planned inference, scoring
# on a THIRD stream 122-124 / 124-125 (a plan keeps the inference queue full, and three saturated queues slow each other
# by more than they overlap: the 96->96 convolution 91 -> 155 us); planned inference, scoring on the TABLE BUILDER's
# stream (two queues: the scorer takes its turns with the next frame's tables) 135.6-137.5 / 127.2-127.7 -- shipped;
# per-operator inference (launch gaps for the scorer to fill), third stream 133-136 / 127-129.
# LIDAL_SCORE_OVERLAP=0: after the inference; LIDAL_SCORE_STREAM=tables|third: force the stream.
_OVERLAP = _os.environ.get('LIDAL_SCORE_OVERLAP', '1')
_STREAM = _os.environ.get('LIDAL_SCORE_STREAM', 'auto')


def _overlap_wanted(model):
    return _OVERLAP != '0'


def _score_stream(model, dev, prefetch):
    from .. import backend as B
    from ..network import plan
    which = _STREAM
    if which == 'auto':
        which = 'tables' if (prefetch and plan.ENABLED and _backbone(model) is not None) else 'third'
    if which == 'tables':
        from ..network.geometry import _state
        return _state(dev)['stream']
    return B.side_stream(dev, 3)


class _Inference:
    """infer_frame over this rank's frames in a fixed order, the coordinate tables of the NEXT frame built on a
    second stream beside the forward pass of the current one (network/geometry.py): a frame's forward then never
    waits for the host to learn the sizes of its own maps."""

    def __init__(self, model, by_id, order, inf_reps, autocast, prefetch):
        self.model, self.by_id, self.order = model, by_id, list(order)
        self.inf_reps, self.autocast = inf_reps, autocast
        self.pos = 0
        self.pf = self.g = None
        net = _backbone(model)
        # (tables ahead of the features only for the networks whose coordinate work the package knows; any other model
        # with the (logits, feat) contract -- wrapped, compiled, custom -- runs with its tables built in line)
        if prefetch and self.order and net is not None:
            from ..network import GeometryPrefetcher
            self.pf = GeometryPrefetcher(net, device=by_id[self.order[0]]['coords'].device)
            self.g = self.pf.submit(by_id[self.order[0]]['coords'], grad=False)

    def __call__(self, f):
        assert f == self.order[self.pos], 'frames are inferred in the announced order'
        d = self.by_id[f]
        prob, _ = infer_frame(self.model, d['coords'], d['feats'], d['inverse'], self.inf_reps,
                              autocast=self.autocast, geometry=self.g)
        self.pos += 1
        if self.pf is not None:
            if self.pos < len(self.order):
                self.g = self.pf.submit(self.by_id[self.order[self.pos]]['coords'], grad=False)
            else:                       # the last frame: fence the tables still held and hand them to the device state
                self.g = None
                self.pf.close()
        return prob


def _backbone(model):
    """The SPVCNN / MinkUNet behind `model` (itself, or the `.module` of a DistributedDataParallel-style wrapper), or
    None."""
    from ..network import SPVCNN, MinkUNet
    for m in (model, getattr(model, 'module', None)):
        if isinstance(m, (SPVCNN, MinkUNet)):
            return m
    return None


def _num_classes(model):
    """Classes of the model's output, known BEFORE any inference (the halo exchange sizes its receive buffers from it):
    the attribute the package's networks carry, else the width of a `classifier` head, else SemanticKITTI's 19."""
    for m in (model, getattr(model, 'module', None)):
        if m is None:
            continue
        n = getattr(m, 'num_classes', None)
        if isinstance(n, int) and n > 0:
            return n
        head = getattr(m, 'classifier', None)
        if head is not None:
            last = [c for c in head.modules() if hasattr(c, 'out_features')]
            if last:
                return int(last[-1].out_features)
    return 19


def score_sequence(model, local_frames, first_frame, n_total, nei_num=24, dis_thresh=0.1,
                   inf_reps=8, autocast=False, group=None, exchange='halo', prefetch=True, overlap=True):
    """local_frames: list of dicts for frames first_frame, first_frame+1, ... owned by this rank,
    each with device tensors coords (i32 [N,4]), feats (f32 [N,4]), inverse (i64 [reps*P]),
    world (f64 [P,3]), sv_ptr / sv_idx (CSR of the supervoxels).
    Returns a list (one entry per local frame) of (sv_interds f32 [S], sv_interes f32 [S],
    sv_centers f32 [S,3]) device tensors.
    exchange: 'halo' (default) -- every rank receives only the frames its block reads (its neighbours'
    edge frames + the wrap-rule frames at the ends of the sequence), point to point; the frames other
    ranks read are inferred first and travel under the inference of the rest; 'allgather' -- every frame to every rank
    (one padded all_gather_into_tensor per array).  Same scores bit for bit.
    prefetch: build each frame's coordinate tables one frame ahead on a second stream (same tables).
    overlap (one rank; LIDAL_SCORE_OVERLAP): a frame is scored as soon as the last frame of its window has been
    inferred, BESIDE the inference of the frames that follow -- on the table builder's stream when the inference is a
    launch plan, on a third stream otherwise (_score_stream) (the reference scores the frames of a sequence concurrently too,
    score/sv_level/LiDAL.py:204-206 `Pool(24)`); the scorer is library kernels only.  Same scores bit for bit."""
    n_class = _num_classes(model)
    dev = local_frames[0]['world'].device if local_frames else None
    if (overlap and _overlap_wanted(model) and not is_sharded(group) and local_frames and len(local_frames) == n_total
            and first_frame == 0):
        return _score_overlapped(model, local_frames, n_total, nei_num, dis_thresh, inf_reps, autocast, prefetch)
    if exchange == 'allgather' or not is_sharded(group):
        probs, worlds = {}, {}
        by_id = {first_frame + s: d for s, d in enumerate(local_frames)}
        infer = _Inference(model, by_id, list(by_id), inf_reps, autocast, prefetch)
        for f, d in by_id.items():
            probs[f] = infer(f)
            worlds[f] = d['world']
            n_class = probs[f].shape[1]
        all_prob = gather_frames(probs, n_total, (n_class,), torch.float32, group=group, device=dev)
        all_world = gather_frames(worlds, n_total, (3,), torch.float64, group=group, device=dev)
        bank = FrameBank(dis_thresh)
        for w, p in zip(all_world, all_prob):
            bank.add(w, p)
    else:
        assert exchange == 'halo', exchange
        if dev is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        by_id = {first_frame + s: d for s, d in enumerate(local_frames)}
        hx = HaloExchange(n_total, nei_num, {f: d['world'].shape[0] for f, d in by_id.items()}, group=group,
                          device=dev)
        worlds = {f: d['world'] for f, d in by_id.items()}
        hx.exchange('world', (3,), torch.float64, worlds)
        probs = {}
        first = [f for f in hx.exports if f in by_id]
        rest = [f for f in by_id if f not in set(first)]
        infer = _Inference(model, by_id, first + rest, inf_reps, autocast, prefetch)
        for f in first:                 # the frames other ranks read: first, so that their transfer ...
            probs[f] = infer(f)
        hx.exchange('prob', (n_class,), torch.float32, probs)
        for f in rest:                  # ... runs under the inference of the rest
            probs[f] = infer(f)
        have = hx.finish({'world': worlds, 'prob': probs})
        bank = FrameBank(dis_thresh, n_frames=n_total)
        for f in sorted(have['world']):
            bank.add(have['world'][f], have['prob'][f], frame_id=f)
    return [score_frame(bank, first_frame + s, d['sv_ptr'], d['sv_idx'], nei_num)
            for s, d in enumerate(local_frames)]


def _score_overlapped(model, frames, n_total, nei_num, dis_thresh, inf_reps, autocast, prefetch):
    """One rank, the whole sequence: inference in frame order on the current stream; frame i's score is queued on a
    third stream the moment every frame of its window (interframe.neighbour_ids) has been inferred."""
    from .. import backend as B
    from .interframe import neighbour_ids
    dev = frames[0]['world'].device
    main = torch.cuda.current_stream(dev)
    side = _score_stream(model, dev, prefetch)
    by_id = dict(enumerate(frames))
    infer = _Inference(model, by_id, list(by_id), inf_reps, autocast, prefetch)
    bank = FrameBank(dis_thresh, n_frames=n_total)
    if any(not 0 <= j < n_total for i in range(n_total) for j in neighbour_ids(i, n_total, nei_num)):
        raise RuntimeError('lidal_amd: a sequence of %d frames is shorter than its neighbour window (nei_num %d; the '
                           'reference, score/sv_level/LiDAL.py:41-42, fails on it too)' % (n_total, nei_num))
    last = {i: max([i] + neighbour_ids(i, n_total, nei_num)) for i in range(n_total)}
    ready_at = {}
    for i, f in last.items():
        ready_at.setdefault(f, []).append(i)
    out = [None] * n_total
    for f in range(n_total):
        d = by_id[f]
        bank.add(d['world'], infer(f), frame_id=f)
        if f in ready_at:
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                for i in ready_at[f]:
                    out[i] = score_frame(bank, i, by_id[i]['sv_ptr'], by_id[i]['sv_idx'], nei_num)
                    for t in out[i]:            # allocated on the third stream, read by the caller on this one
                        t.record_stream(main)
    main.wait_stream(side)
    return out


def collect_sequence(local_scores, local_sv_ids, local_sv_ptrs, first_frame, n_total, group=None):
    """The return leg of a sequence (LiDAL.py:206-207, `re_values = pool.map(worker_func, ids)`):
    local_scores = score_sequence()'s list for this rank's frames, local_sv_ids = per frame the
    global supervoxel ids (i64 [S], the pickle's `sv_id`), local_sv_ptrs = per frame the CSR
    pointer (pnums = its differences, LiDAL.py:94).  On rank 0 returns, for every frame of the
    sequence in order, worker_func's tuple (sv_id i64 [S], sv_interds f32 [S], sv_interes f32 [S],
    sv_pnums i64 [S], sv_centers f32 [S,3]) as numpy arrays; None on the other ranks.
    Everything rides in ONE f64 [S,7] block per frame (f32 values, ids and counts are exact in f64)."""
    packed = {}
    for s, (sc, ids, ptr) in enumerate(zip(local_scores, local_sv_ids, local_sv_ptrs)):
        sv_d, sv_e, sv_c = sc
        ids = torch.as_tensor(np.asarray(ids), dtype=torch.float64, device=sv_d.device)
        pn = (ptr[1:] - ptr[:-1]).to(device=sv_d.device, dtype=torch.float64)
        packed[first_frame + s] = torch.cat(
            [ids[:, None], sv_d.double()[:, None], sv_e.double()[:, None], pn[:, None],
             sv_c.double()], dim=1).contiguous()
    if is_sharded(group):
        dev = local_scores[0][0].device if local_scores else None
        got = gather_frames(packed, n_total, (7,), torch.float64, group=group, device=dev)
        if dist.get_rank(group) != 0:
            return None
    else:
        got = [packed[f] for f in range(n_total)]
    out = []
    for g in got:
        g = g.cpu().numpy()
        out.append((g[:, 0].astype(np.int64), g[:, 1].astype(np.float32), g[:, 2].astype(np.float32),
                    g[:, 3].astype(np.int64), g[:, 4:7].astype(np.float32)))
    return out


class ScoreBoard:
    """Rank 0's global per-supervoxel arrays (LiDAL.py:167-182) + the scatter of each sequence's
    results into them (:208-218) + the selection call (:225-325)."""

    def __init__(self, n_sv, sv_pnums=None, sv_centers=None):
        self.sv_interds = np.zeros(n_sv, dtype=np.float32)
        self.sv_interes = np.zeros(n_sv, dtype=np.float32)
        self.sv_pre = sv_pnums is not None          # cached stats from an earlier round (:173-177)
        self.sv_pnums = np.asarray(sv_pnums) if self.sv_pre else np.zeros(n_sv, dtype=int)
        self.sv_centers = (np.asarray(sv_centers) if self.sv_pre
                           else np.zeros((n_sv, 3), dtype=np.float32))

    def add_sequence(self, seq_index, frames):
        """frames: collect_sequence()'s list for sequence number `seq_index` of the split."""
        for sv_id, d, e, n, c in frames:
            self.sv_interds[sv_id] = d
            self.sv_interes[sv_id] = e
            if not self.sv_pre:
                self.sv_pnums[sv_id] = n
                # offset that keeps supervoxels of different sequences > 5 m apart (:218)
                self.sv_centers[sv_id] = c + seq_index * 1000.0

    def select(self, sv_flags, train_point_num, sv_dis_thresh=5.0):
        return select(sv_flags, self.sv_interds, self.sv_interes, self.sv_pnums, self.sv_centers,
                      train_point_num, sv_dis_thresh)
