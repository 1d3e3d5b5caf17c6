"""Host-side (Python) profile of prob_inference on one frame (8 views): where launch-bound time goes."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN
from lidal_amd.score.prob_inference import infer_frame
dev = torch.device('cuda:0')
model = SPVCNN(19).to(dev).eval()
f = synth.make_sequence(1, n_points=120000, seed=7122, start=0, total=1)[0]
sb = synth.make_score_batch(f['points'], f['intensity'], np.random.default_rng(1), inf_reps=8)
c = torch.from_numpy(sb['coords_v_b']).to(dev); x = torch.from_numpy(sb['feats_v_b']).to(dev)
inv = torch.from_numpy(sb['inverse_indices_b']).to(dev)
for _ in range(3): infer_frame(model, c, x, inv, 8, autocast=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): infer_frame(model, c, x, inv, 8, autocast=True)
torch.cuda.synchronize(); print('ms/frame', (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): infer_frame(model, c, x, inv, 8, autocast=True)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:9000])
