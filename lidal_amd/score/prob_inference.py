"""One step of /root/reference/score/prob_inference.py:91-113 kept on the GPU.

The reference copies the logits of the 8 collated views to the host (61 MB per frame), gathers
voxel -> point with the inverse indices, soft-maxes, averages the views and arg-maxes in numpy.
Here the model forward and one fused kernel (lidal_view_mean_softmax) do all of it in HBM; only
what the scorer needs next ([P, C] probabilities) stays resident.
"""
import torch

from .. import SparseTensor
from .. import backend as B

__all__ = ['infer_frame', 'view_mean_softmax']


def view_mean_softmax(logits, inverse_indices, inf_reps):
    """logits f32 [sum_v N_v, C]; inverse_indices i64 [inf_reps * P] (collated, offset per view as
    dataset/sk_dataset.py:214-217) -> (prob f32 [P, C], pred i64 [P])."""
    B.require_gpu(logits, inverse_indices)
    logits = logits.contiguous().float()
    inverse_indices = inverse_indices.contiguous()
    assert inverse_indices.dtype == torch.int64
    assert inverse_indices.numel() % inf_reps == 0
    p = inverse_indices.numel() // inf_reps
    c = logits.shape[1]
    prob = torch.empty((p, c), dtype=torch.float32, device=logits.device)
    pred = torch.empty(p, dtype=torch.int64, device=logits.device)
    B.check(B.lib().lidal_view_mean_softmax(B.ptr(logits), B.ptr(inverse_indices), inf_reps, p, c,
                                            B.ptr(prob), B.ptr(pred), B.stream()),
            'view_mean_softmax')
    return prob, pred


@torch.no_grad()
def infer_frame(model, coords_v_b, feats_v_b, inverse_indices_b, inf_reps=8, autocast=False, return_feat=False,
                geometry=None):
    """model.eval() forward over the `inf_reps` augmented views of ONE frame, then the fused
    voxel->point gather + softmax + view mean + argmax.  Returns (prob [P,C], pred [P]).
    return_feat (prob_inference.py:103-105,116-118: `outfeat`, saved when r_id == 0 or the metric is
    ReDAL / CSET): also the [P, 96] feature of every point, the view mean of feat[inverse_indices] as the
    reference computes it -- (prob, pred, feat)."""
    x = SparseTensor(feats_v_b, coords_v_b)
    if geometry is not None:        # the frame's coordinate tables, built ahead (lidal_amd.network.GeometryPrefetcher)
        x.geometry = geometry
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        logits, feat = model(x)
    prob, pred = view_mean_softmax(logits, inverse_indices_b, inf_reps)
    if not return_feat:
        return prob, pred
    p = inverse_indices_b.numel() // inf_reps
    feat_p = feat.float()[inverse_indices_b].reshape(inf_reps, p, feat.shape[1]).mean(0)
    return prob, pred, feat_p
