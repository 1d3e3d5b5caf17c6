import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lidal_amd
from lidal_amd import backend as B, synth
from lidal_amd.network import SPVCNN, MinkUNet, plan
from lidal_amd.score.prob_inference import infer_frame
dev = 'cuda'
for name, cls in (('minkunet', MinkUNet), ('spvcnn', SPVCNN)):
    torch.manual_seed(4)
    model = cls(19).to(dev).eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
    seq = synth.make_sequence(1, n_points=9000, seed=30)[0]
    sb = synth.make_score_batch(seq['points'], seq['intensity'], np.random.default_rng(0), inf_reps=8)
    c, f, inv = (torch.from_numpy(sb[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'inverse_indices_b'))
    res = {}
    for split in (False, True):
        B.SPLIT_F32 = split
        for planned in (False, True):
            plan.ENABLED = planned
            with torch.no_grad():
                logits, feat = model(lidal_amd.SparseTensor(f, c))
            res[(split, planned)] = (logits.float().clone(), feat.float().clone())
    ref = res[(False, False)]
    for k, v in res.items():
        print(name, 'split=%s planned=%s' % k, 'logits maxdiff vs exact per-op: %.3e' % (v[0] - ref[0]).abs().max().item(),
              'feat: %.3e (scale %.2f)' % ((v[1] - ref[1]).abs().max().item(), ref[1].abs().max().item()))
