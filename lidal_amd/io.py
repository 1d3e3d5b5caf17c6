"""On-disk formats shared with an existing LiDAL `Processing_files/` tree and the reference's
checkpoints.  Plain host I/O: the formats are numpy .npy, pickles of
numpy arrays and torch.save dicts, so files written here are readable by the reference scripts and
vice versa.

  prob_map/.../<frame>.npy     f32 [P, C]    score/prob_inference.py:129 -> LiDAL.py:45-49
  pred/.../<frame>.npy         i64 [P]       score/prob_inference.py:130
  super_voxel/.../<frame>.pickle (sv_id i64 [S], sv2point list of i64 arrays)
                                              dataset/prepare_supervoxel_kmeans_sk.py:62-74
  sv_flag/.../<frame>.npy      i64 [S] in {0,1,2}   LiDAL.py:328-330
  super_voxel/KMeans/sv_pnums.npy, sv_centers.npy   i64 [sum S]; f32 [sum S, 3] with the
                               +1000 * sequence-index offset      LiDAL.py:173-177,220-222
  <dir>/current.pt             {'model_state_dict', 'iteration', 'ep_id'}   train.py:151-155
"""
import os
import pickle

import numpy as np
import torch

__all__ = ['save_prob_pred', 'load_prob', 'load_supervoxels', 'save_supervoxels', 'load_sv_flag',
           'save_sv_flag', 'load_sv_stats', 'save_sv_stats', 'save_checkpoint', 'load_checkpoint']


def _mkdir_for(path):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)


def save_prob_pred(prob_path, pred_path, prob, pred):
    """prob f32 [P,C] / pred i64 [P] (tensors or arrays) -> the two .npy files of prob_inference."""
    prob = prob.detach().cpu().numpy() if torch.is_tensor(prob) else np.asarray(prob)
    pred = pred.detach().cpu().numpy() if torch.is_tensor(pred) else np.asarray(pred)
    _mkdir_for(prob_path), _mkdir_for(pred_path)
    np.save(prob_path, prob.astype(np.float32, copy=False))
    np.save(pred_path, pred.astype(np.int64, copy=False))


def load_prob(path, device=None):
    prob = np.load(path)
    assert prob.dtype == np.float32 and prob.ndim == 2, (prob.dtype, prob.shape)
    t = torch.from_numpy(prob)
    return t.to(device) if device is not None else t


def load_supervoxels(path):
    """-> (sv_id i64 [S], sv2point list of i64 index arrays) exactly as LiDAL.py:84-85 reads it."""
    with open(path, 'rb') as f:
        sv_id, sv2point = pickle.load(f)
    return np.asarray(sv_id), [np.asarray(p) for p in sv2point]


def save_supervoxels(path, sv_id, sv2point):
    _mkdir_for(path)
    with open(path, 'wb') as f:
        pickle.dump((np.asarray(sv_id), [np.asarray(p) for p in sv2point]), f)


def load_sv_flag(path):
    return np.load(path)


def save_sv_flag(path, flags):
    _mkdir_for(path)
    np.save(path, np.asarray(flags))


def load_sv_stats(pnums_path, centers_path):
    """The cached per-supervoxel statistics of LiDAL.py:173-177: (sv_pnums i64 [N], sv_centers f32
    [N,3]) -- feed them to score.ScoreBoard(n, sv_pnums, sv_centers) (the reference's `sv_pre`)."""
    pn, ce = np.load(pnums_path), np.load(centers_path)
    assert pn.ndim == 1 and ce.shape == (pn.shape[0], 3), (pn.shape, ce.shape)
    return pn, ce


def save_sv_stats(pnums_path, centers_path, sv_pnums, sv_centers):
    """LiDAL.py:220-222 (written once, by the first scoring round)."""
    _mkdir_for(pnums_path), _mkdir_for(centers_path)
    np.save(pnums_path, np.asarray(sv_pnums))
    np.save(centers_path, np.asarray(sv_centers))


def save_checkpoint(directory, model, iteration, ep_id):
    """train.py:150-155 (rank 0 only in the reference); unwraps DistributedDataParallel."""
    os.makedirs(directory, exist_ok=True)
    module = model.module if hasattr(model, 'module') else model
    torch.save({'model_state_dict': module.state_dict(), 'iteration': iteration, 'ep_id': ep_id},
               os.path.join(directory, 'current.pt'))


def load_checkpoint(path, model, strict=True, map_location='cpu'):
    """train.py:63-72 / prob_inference.py:65-71: load a reference-format checkpoint (keys may carry
    DDP's 'module.' prefix); returns (iteration, ep_id)."""
    ckpt = torch.load(path, map_location=map_location)
    sd = {k[len('module.'):] if k.startswith('module.') else k: v
          for k, v in ckpt['model_state_dict'].items()}
    (model.module if hasattr(model, 'module') else model).load_state_dict(sd, strict=strict)
    return ckpt.get('iteration', 0), ckpt.get('ep_id', 0)
