"""CPU restatement of the LiDAL inter-frame scorer and selection.  TEST INFRASTRUCTURE.

Follows /root/reference/score/sv_level/LiDAL.py:27-103 (worker_func) and :225-330 (AL + SL
selection) on in-memory arrays instead of files.  PINNED against the reference's own
worker_func and __main__ (tests/golden/make_golden.py -> tests/golden/scoring_small.npz).
"""
import numpy as np
from scipy.special import kl_div
from scipy.stats import entropy
from sklearn.neighbors import KDTree


def neighbour_ids(i, n_frames, nei_num):
    """LiDAL.py:41-42 -- nei_num/2 frames before and after with the reference's wrap rules."""
    half = int(nei_num / 2)
    ids = [(i - o - 1) if (i - o - 1) >= 0 else (half + o + 1) for o in range(half)]
    ids += [(i + o + 1) if (i + o + 1) <= (n_frames - 1) else (n_frames - 2 - half - o)
            for o in range(half)]
    return ids


def score_frame(i, probs, worlds, sv2point, nei_num=24, dis_thresh=0.1, trees=None,
                return_points=False):
    """LiDAL.py:27-103 for frame i.  probs: list of f32 [P_f, C]; worlds: list of f64 [P_f, 3]
    (the data held by the reference's pickled KDTrees); sv2point: list of index arrays.
    Returns sv_interds f32 [S], sv_interes f32 [S], sv_pnums i64 [S], sv_centers f32 [S,3]
    (+ interd_points f64 [P], intere_points f32 [P] when return_points)."""
    nei = neighbour_ids(i, len(probs), nei_num)
    query_prob = probs[i]
    query_points = np.asarray(worlds[i])
    map_count = np.ones(query_prob.shape[0])
    interd_points = np.zeros(query_points.shape[0])
    sum_prob = query_prob.copy()
    epsilon = 0.00001
    for n in nei:
        tree = trees[n] if trees is not None else KDTree(worlds[n])
        n_prob = probs[n]
        dists, nearest = tree.query(query_points, k=1, return_distance=True, dualtree=False,
                                    breadth_first=False)
        dists = dists.squeeze()
        nearest = nearest.squeeze()
        m = dists <= dis_thresh
        sum_prob[m] += n_prob[nearest][m]
        interd_points[m] += np.sum(kl_div(query_prob[m] + epsilon, n_prob[nearest][m] + epsilon),
                                   axis=1)
        map_count[m] += 1
    sum_prob /= np.expand_dims(map_count, 1)
    intere_points = entropy(sum_prob, axis=1)
    map_count = map_count - 1
    mm = map_count > 0
    interd_points[mm] /= map_count[mm]
    S = len(sv2point)
    sv_interds = np.zeros(S, dtype=np.float32)
    sv_interes = np.zeros(S, dtype=np.float32)
    sv_pnums = np.zeros(S, dtype=int)
    sv_centers = np.zeros((S, 3), dtype=np.float32)
    for s, p_ids in enumerate(sv2point):
        sv_pnums[s] = len(p_ids)
        sv_centers[s] = query_points[p_ids].mean(0)
        sv_interds[s] = interd_points[p_ids].mean()
        sv_interes[s] = intere_points[p_ids].mean()
    if return_points:
        return sv_interds, sv_interes, sv_pnums, sv_centers, interd_points, intere_points
    return sv_interds, sv_interes, sv_pnums, sv_centers


def select(sv_flags, sv_interds, sv_interes, sv_pnums, sv_centers, train_point_num,
           sv_dis_thresh=5.0):
    """LiDAL.py:225-325.  sv_flags int array in {0,1,2}; returns the new flags.
    NB: iterates a Python set and breaks on the first hit, exactly as the reference does, so
    the outcome depends on CPython set iteration order -- kept on purpose."""
    sv_flags = np.array(sv_flags).astype(int)
    unlabeled_ids = np.where(sv_flags == 0)[0]
    unlabeled_interds = sv_interds[unlabeled_ids]
    sorted_ids = np.argsort(unlabeled_interds)
    added_ids = set()
    point_limit = round(0.01 * train_point_num)
    for idx in reversed(sorted_ids):
        sv_id = unlabeled_ids[idx]
        sv_c = sv_centers[sv_id]
        flag = True
        for l_sv_id in added_ids:
            l_sv_c = sv_centers[l_sv_id]
            dist = np.sqrt(np.square(sv_c - l_sv_c).sum())
            if dist < sv_dis_thresh:
                flag = False
                if sv_interes[l_sv_id] < sv_interes[sv_id]:
                    sv_flags[sv_id] = 1
                    sv_flags[l_sv_id] = 0
                    added_ids.add(sv_id)
                    added_ids.remove(l_sv_id)
                    point_limit = point_limit + sv_pnums[l_sv_id] - sv_pnums[sv_id]
                break
        if flag:
            point_limit -= sv_pnums[sv_id]
            if point_limit < 0:
                break
            sv_flags[sv_id] = 1
            added_ids.add(sv_id)

    unlabeled_ids = np.where(sv_flags == 0)[0]
    unlabeled_interds = sv_interds[unlabeled_ids]
    sorted_ids = np.argsort(unlabeled_interds)
    sv_flags[sv_flags == 2] = 0
    added_ids = set()
    point_limit = round(0.01 * train_point_num)
    for idx in sorted_ids:
        if unlabeled_interds[idx] == 0:
            continue
        sv_id = unlabeled_ids[idx]
        sv_c = sv_centers[sv_id]
        flag = True
        for l_sv_id in added_ids:
            l_sv_c = sv_centers[l_sv_id]
            dist = np.sqrt(np.square(sv_c - l_sv_c).sum())
            if dist < sv_dis_thresh:
                flag = False
                if sv_interes[l_sv_id] > sv_interes[sv_id]:
                    sv_flags[sv_id] = 2
                    sv_flags[l_sv_id] = 0
                    added_ids.add(sv_id)
                    added_ids.remove(l_sv_id)
                    point_limit = point_limit + sv_pnums[l_sv_id] - sv_pnums[sv_id]
                break
        if flag:
            point_limit -= sv_pnums[sv_id]
            if point_limit < 0:
                break
            sv_flags[sv_id] = 2
            added_ids.add(sv_id)
    return sv_flags
