"""Where the host time of a planned single-scan step goes: cProfile of 30 steps (tables prefetched)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN, GeometryPrefetcher  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402

dev = 'cuda'
from lidal_amd import backend as B  # noqa: E402
B.bind_cpus_near(0)
b = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '1')), n_points=120000, seed=7122)
coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)
pf = GeometryPrefetcher(model)
g = pf.submit(coords)
for _ in range(5):
    train_step(model, opt, feats, coords, labels, autocast=True, geometry=g)
    g = pf.submit(coords)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    train_step(model, opt, feats, coords, labels, autocast=True, geometry=g)
    g = pf.submit(coords)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(38)
st.sort_stats('tottime').print_stats(22)
