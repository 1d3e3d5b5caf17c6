"""The secondary metric's pipeline (score_sequence on one rank: inference on the main stream, the next frame's tables
on a second, scoring on a third) for `rocprofv3 --kernel-trace`: the timed pass is bracketed by two launches of
transpose_f32_kernel (markers scripts/exp/score_timeline_read.py cuts the trace at).
    python scripts/exp/score_timeline.py [frames] [nei]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lidal_amd import backend as B, synth                       # noqa: E402
from lidal_amd.network import SPVCNN                            # noqa: E402
from lidal_amd.score import interframe, score_sequence          # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nei = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
torch.manual_seed(7122)
model = SPVCNN(19).to(dev).eval()
frames = synth.make_sequence(per, n_points=120000, seed=7122, start=0, total=per)
rng = np.random.default_rng([7122, 99, 0])
dev_frames = []
for f in frames:
    sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=8)
    ptr, idx, _ = interframe.sv_csr(f['sv2point'], dev)
    dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(dev), 'feats': torch.from_numpy(sb['feats_v_b']).to(dev),
                       'inverse': torch.from_numpy(sb['inverse_indices_b']).to(dev),
                       'world': torch.from_numpy(f['world']).to(dev), 'sv_ptr': ptr, 'sv_idx': idx})
mark_src = torch.zeros((8, 8), device=dev)
mark_dst = torch.zeros((8, 8), device=dev)


def mark():
    B.check(B.lib().lidal_transpose_f32(B.ptr(mark_src), 8, B.ptr(mark_dst), 8, 8, B.stream()), 'mark')


score_sequence(model, dev_frames, 0, per, nei_num=nei, dis_thresh=0.1, inf_reps=8, autocast=True)
torch.cuda.synchronize()
for rep in range(2):
    mark()
    t0 = time.perf_counter()
    out = score_sequence(model, dev_frames, 0, per, nei_num=nei, dis_thresh=0.1, inf_reps=8, autocast=True)
    t_host = time.perf_counter() - t0
    mark()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('rep %d: %.2f ms/frame (%.1f frames/s); host queued everything after %.2f ms/frame' % (rep, dt / per * 1e3, per / dt, t_host / per * 1e3))
