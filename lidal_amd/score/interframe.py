"""Inter-frame divergence and entropy per point and per supervoxel on the GPU
(counterpart of /root/reference/score/sv_level/LiDAL.py:27-103, `worker_func`).

The reference gives every frame to one of 24 CPU workers which loads 25 probability maps and 25
pickled sklearn KD-trees from disk and runs 24 nearest-neighbour queries.  Here every frame's
world-frame points (f64 [P,3]) and probabilities (f32 [P,C]) stay resident in HBM; a uniform
grid with cell = match radius is built ONCE per frame (radix sort by cell + hash table of cell
starts) and re-used by the up to 24 frames that see it as a neighbour; one kernel per query
frame walks its neighbours in the reference's order and accumulates KL / mean probability, then
one kernel reduces per supervoxel.
"""
import ctypes
import os

import numpy as np
import torch

from .. import backend as B

__all__ = ['FrameBank', 'neighbour_ids', 'score_frame']

# Round 6, measured and NOT adopted (scripts/gpu/r6_check2.sh, 32 frames of 120 k points, nei 10): the queries of a frame taken
# in the cell order of its own grid (lidal_interframe_score_ordered: a wave's queries then sit in a handful of neighbouring
# cells) -- scorer kernels 618 us per frame in scan order, 623 in cell order; frames/s 75.2 / 74.9.  A LiDAR scan is already
# ordered along its rings, consecutive points ARE neighbours; what the match kernel waits for is the dependent chain bitmap
# word -> slot -> record of each probed cell, not the sectors.  LIDAL_SCORE_CELL_ORDER=1 switches it on (same scores bit for bit).
CELL_ORDER = os.environ.get('LIDAL_SCORE_CELL_ORDER', '0') == '1'


def neighbour_ids(i, n_frames, nei_num):
    """LiDAL.py:41-42: nei_num/2 frames before and after i; ids falling off either end of the
    sequence are replaced by frames further on the other side."""
    half = int(nei_num / 2)
    before = [(i - o - 1) if (i - o - 1) >= 0 else (half + o + 1) for o in range(half)]
    after = [(i + o + 1) if (i + o + 1) <= (n_frames - 1) else (n_frames - 2 - half - o)
             for o in range(half)]
    return before + after


class FrameBank:
    """Device-resident frames of one sequence + lazily built nearest-neighbour grids.  Frames are keyed
    by their id in the sequence; a bank may hold only SOME of them (a rank's block and its halo,
    score/sharding.py HaloExchange): pass `n_frames` (the sequence length, which the neighbour rule
    needs) and `frame_id` to add()."""

    CELL = float(__import__('os').environ.get('LIDAL_GRID_CELL', '2'))      # grid cell in units of the match radius

    def __init__(self, dis_thresh=0.1, n_frames=None):
        self.dis_thresh = float(dis_thresh)
        self.n_frames = n_frames
        self.world = {}       # frame id -> f64 [P,3]
        self.prob = {}        # frame id -> f32 [P,C]
        self._grid = {}

    def __len__(self):
        return self.n_frames if self.n_frames is not None else len(self.world)

    def add(self, world, prob, frame_id=None):
        B.require_gpu(world, prob)
        assert world.dtype == torch.float64 and world.shape[1] == 3
        assert prob.dtype == torch.float32 and prob.shape[0] == world.shape[0]
        f = len(self.world) if frame_id is None else int(frame_id)
        assert f not in self.world
        self.world[f] = world.contiguous()
        self.prob[f] = prob.contiguous()
        self._grid[f] = None

    def grid(self, i):
        if self._grid[i] is None:
            pts = self.world[i]
            p = pts.shape[0]
            nbytes = B.lib().lidal_nn_grid_bytes(p)
            ws_bytes = B.lib().lidal_nn_grid_workspace_bytes(p)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pts.device)
            # cells of twice the match radius: a query then probes at most 8 cells instead of 27 (same matches)
            B.check(B.lib().lidal_nn_grid_build(B.ptr(pts), p, self.CELL * self.dis_thresh, B.ptr(buf), nbytes,
                                                B.ptr(ws), ws_bytes, B.stream()), 'nn_grid_build')
            self._grid[i] = buf
        return self._grid[i]


def _ptr_array(ctype, tensors):
    return (ctype * len(tensors))(*[t.data_ptr() for t in tensors])


def score_points(bank, i, nei_num=24):
    """Per-point (interd f64 [P], intere f32 [P], map_count i32 [P]) of frame i."""
    nei = neighbour_ids(i, len(bank), nei_num)
    q_pts, q_prob = bank.world[i], bank.prob[i]
    p, c = q_prob.shape
    dev = q_pts.device
    interd = torch.empty(p, dtype=torch.float64, device=dev)
    intere = torch.empty(p, dtype=torch.float32, device=dev)
    count = torch.empty(p, dtype=torch.int32, device=dev)
    grids = [bank.grid(n) for n in nei]
    g_arr = _ptr_array(ctypes.c_void_p, grids)
    p_arr = _ptr_array(ctypes.c_void_p, [bank.world[n] for n in nei])
    f_arr = _ptr_array(ctypes.c_void_p, [bank.prob[n] for n in nei])
    n_arr = (ctypes.c_int64 * len(nei))(*[bank.world[n].shape[0] for n in nei])
    for n in nei:
        assert bank.prob[n].shape[1] == c
    ws_bytes = B.lib().lidal_interframe_workspace_bytes(p, len(nei))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    q_grid = bank.grid(i) if CELL_ORDER else None          # (see CELL_ORDER)
    B.check(B.lib().lidal_interframe_score_ordered(B.ptr(q_pts), B.ptr(q_prob), p, c, g_arr, p_arr, f_arr,
                                                   n_arr, len(nei), bank.dis_thresh, B.ptr(interd),
                                                   B.ptr(intere), B.ptr(count), B.ptr(ws), ws_bytes,
                                                   B.ptr(q_grid), B.stream()),
            'interframe_score')
    return interd, intere, count


def sv_csr(sv2point, device):
    """list of index arrays (the reference's sv2point) -> (ptr i64 [S+1], idx i64 [sum])."""
    lens = np.array([len(s) for s in sv2point], dtype=np.int64)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = (np.concatenate([np.asarray(s, dtype=np.int64) for s in sv2point])
           if len(sv2point) else np.zeros(0, np.int64))
    return torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device), lens


def score_frame(bank, i, sv_ptr, sv_idx, nei_num=24):
    """LiDAL.py:59-98 for frame i: returns (sv_interds f32 [S], sv_interes f32 [S],
    sv_centers f32 [S,3]) as device tensors."""
    interd, intere, _ = score_points(bank, i, nei_num)
    s = sv_ptr.numel() - 1
    dev = interd.device
    sv_d = torch.empty(s, dtype=torch.float32, device=dev)
    sv_e = torch.empty(s, dtype=torch.float32, device=dev)
    sv_c = torch.empty((s, 3), dtype=torch.float32, device=dev)
    B.check(B.lib().lidal_supervoxel_reduce(B.ptr(interd), B.ptr(intere), B.ptr(bank.world[i]),
                                            B.ptr(sv_ptr), B.ptr(sv_idx), s, B.ptr(sv_d),
                                            B.ptr(sv_e), B.ptr(sv_c), B.stream()),
            'supervoxel_reduce')
    return sv_d, sv_e, sv_c
