"""Phase timing of the secondary metric (prob_inference 8 views + LiDAL inter-frame scoring) on one
GPU; run plain for host-side phase times, or under `rocprofv3 --kernel-trace --stats` for kernels.
    python scripts/profile_scoring.py [frames] [points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidal_amd import synth                                     # noqa: E402
from lidal_amd.network import SPVCNN                            # noqa: E402
from lidal_amd.score import interframe                          # noqa: E402
from lidal_amd.score.interframe import FrameBank, score_frame   # noqa: E402
from lidal_amd.score.prob_inference import infer_frame          # noqa: E402


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    points = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
    dev = torch.device('cuda:0')
    torch.manual_seed(7122)
    model = SPVCNN(19).to(dev).eval()
    frames = synth.make_sequence(per, n_points=points, seed=7122, start=0, total=per)
    rng = np.random.default_rng([7122, 99, 0])
    dev_frames = []
    for f in frames:
        sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=8)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], dev)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(dev),
                           'feats': torch.from_numpy(sb['feats_v_b']).to(dev),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(dev),
                           'world': torch.from_numpy(f['world']).to(dev), 'sv_ptr': ptr, 'sv_idx': idx})
    print('voxels per frame (8 views):', dev_frames[0]['coords'].shape[0])

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    for rep in range(2):
        t0 = sync()
        probs = [infer_frame(model, d['coords'], d['feats'], d['inverse'], 8, autocast=os.environ.get('SCORE_DTYPE', 'bf16') == 'bf16')[0]
                 for d in dev_frames]
        t1 = sync()
        bank = FrameBank(0.1)
        for d, p in zip(dev_frames, probs):
            bank.add(d['world'], p)
        t2 = sync()
        out = [score_frame(bank, s, d['sv_ptr'], d['sv_idx'], 10) for s, d in enumerate(dev_frames)]
        t3 = sync()
        print('rep %d: inference %.2f ms/frame, bank %.2f ms/frame, scoring %.2f ms/frame'
              % (rep, (t1 - t0) / per * 1e3, (t2 - t1) / per * 1e3, (t3 - t2) / per * 1e3))
    assert all(torch.isfinite(o[0]).all() for o in out)


if __name__ == '__main__':
    main()
