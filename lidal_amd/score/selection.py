"""Greedy active / pseudo-label selection (counterpart of
/root/reference/score/sv_level/LiDAL.py:225-325).  Deliberately host-side Python on numpy
arrays with the SAME constructs as the reference (np.argsort order, iteration over a Python
`set`, first-hit break): the outcome depends on CPython's set iteration order, so
a re-ordered or parallel version could not reproduce the reference's flags.

Two passes share one routine:
  AL  (flag 1): unlabeled supervoxels by DESCENDING divergence; a candidate within 5 m of an
      already added one replaces it only if its entropy is HIGHER.
  SL  (flag 2): candidates are the supervoxels still unlabeled after AL *before* old flag-2s are
      reset (so last round's pseudo labels are not re-picked), ASCENDING divergence, zeros
      skipped; replacement only if entropy is LOWER.
Each pass stops when 1 % of `train_point_num` points has been spent.
"""
import numpy as np

__all__ = ['select']


def _greedy_pass(order, cand_ids, cand_div, flags, label, prefer_higher_entropy, skip_zero,
                 sv_interes, sv_pnums, sv_centers, budget, radius):
    added = set()
    for idx in order:
        if skip_zero and cand_div[idx] == 0:
            continue
        sv = cand_ids[idx]
        centre = sv_centers[sv]
        free = True
        for other in added:
            dist = np.sqrt(np.square(centre - sv_centers[other]).sum())
            if dist < radius:
                free = False
                wins = (sv_interes[other] < sv_interes[sv] if prefer_higher_entropy
                        else sv_interes[other] > sv_interes[sv])
                if wins:
                    flags[sv] = label
                    flags[other] = 0
                    added.add(sv)
                    added.remove(other)
                    budget = budget + sv_pnums[other] - sv_pnums[sv]
                break
        if free:
            budget -= sv_pnums[sv]
            if budget < 0:
                break
            flags[sv] = label
            added.add(sv)
    return flags


def select(sv_flags, sv_interds, sv_interes, sv_pnums, sv_centers, train_point_num,
           sv_dis_thresh=5.0):
    flags = np.array(sv_flags).astype(int)
    cand = np.where(flags == 0)[0]
    div = sv_interds[cand]
    order = np.argsort(div)
    flags = _greedy_pass(reversed(order), cand, div, flags, 1, True, False, sv_interes, sv_pnums,
                         sv_centers, round(0.01 * train_point_num), sv_dis_thresh)
    cand = np.where(flags == 0)[0]
    div = sv_interds[cand]
    order = np.argsort(div)
    flags[flags == 2] = 0
    flags = _greedy_pass(order, cand, div, flags, 2, False, True, sv_interes, sv_pnums,
                         sv_centers, round(0.01 * train_point_num), sv_dis_thresh)
    return flags
