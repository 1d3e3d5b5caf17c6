"""Host-side (Python) profile of one training step at batch 1: where the launch-bound time goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN
from lidal_amd.train_step import train_step
b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
dev = 'cuda'
coords = torch.from_numpy(b['coords_v_b']).to(dev); feats = torch.from_numpy(b['feats_v_b']).to(dev); labels = torch.from_numpy(b['labels_v_b']).to(dev)
model = SPVCNN(19).to(dev).train(); opt = torch.optim.Adam(model.parameters(), fused=True)
for _ in range(3): train_step(model, opt, feats, coords, labels, autocast=True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): train_step(model, opt, feats, coords, labels, autocast=True)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
