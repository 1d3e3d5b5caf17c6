"""One scoring pass over a sequence, frame-sharded: the GPU counterpart of running
/root/reference/score/prob_inference.py (per-frame 8-view inference, :91-133) followed by
score/sv_level/LiDAL.py:185-218 (per-frame inter-frame divergence / entropy per supervoxel).

Each rank owns a contiguous block of frames (dataset/sk_dataloader.py:196-198); probabilities and
world coordinates are exchanged by one all-gather (replacing the .npy / KD-tree pickle hand-off),
after which every rank scores its own frames against full +-nei windows.
"""
import torch

from .interframe import FrameBank, score_frame
from .prob_inference import infer_frame
from .sharding import gather_frames

__all__ = ['score_sequence']


def score_sequence(model, local_frames, first_frame, n_total, nei_num=24, dis_thresh=0.1,
                   inf_reps=8, autocast=False):
    """local_frames: list of dicts for frames first_frame, first_frame+1, ... owned by this rank,
    each with device tensors coords (i32 [N,4]), feats (f32 [N,4]), inverse (i64 [reps*P]),
    world (f64 [P,3]), sv_ptr / sv_idx (CSR of the supervoxels).
    Returns a list (one entry per local frame) of (sv_interds f32 [S], sv_interes f32 [S],
    sv_centers f32 [S,3]) device tensors."""
    probs, worlds = {}, {}
    for s, d in enumerate(local_frames):
        prob, _ = infer_frame(model, d['coords'], d['feats'], d['inverse'], inf_reps,
                              autocast=autocast)
        probs[first_frame + s] = prob
        worlds[first_frame + s] = d['world']
    all_prob = gather_frames(probs, n_total)
    all_world = gather_frames(worlds, n_total)
    bank = FrameBank(dis_thresh)
    for w, p in zip(all_world, all_prob):
        bank.add(w, p)
    return [score_frame(bank, first_frame + s, d['sv_ptr'], d['sv_idx'], nei_num)
            for s, d in enumerate(local_frames)]
