"""Host time of the sections of a single-scan training step (no synchronisation inside the step): where a
host-bound step spends its Python / launch time.  GPU time of the same step from events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidal_amd import SparseTensor, synth
from lidal_amd.network import SPVCNN, GeometryPrefetcher
from lidal_amd.nn.functional.fused import cross_entropy

dev = 'cuda'
frames = int(os.environ.get('FRAMES', '1'))
b = synth.make_train_batch(n_frames=frames, n_points=120000, seed=7122)
coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)
pf = GeometryPrefetcher(model)
g = pf.submit(coords)
T = {k: [] for k in ('zero_grad', 'forward', 'loss', 'backward', 'adam', 'submit', 'step', 'gpu')}
for it in range(25):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    x = SparseTensor(feats, coords); x.geometry = g
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits, _ = model(x)
    t.append(time.perf_counter())
    loss = cross_entropy(logits, labels, ignore_index=255); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    e1.record()
    g = pf.submit(coords); t.append(time.perf_counter())
    torch.cuda.synchronize()
    if it >= 5:
        for k, a, c in zip(('zero_grad', 'forward', 'loss', 'backward', 'adam', 'submit'), t[:-1], t[1:]):
            T[k].append((c - a) * 1e3)
        T['step'].append((t[-1] - t[0]) * 1e3)
        T['gpu'].append(e0.elapsed_time(e1))
print('frames', frames, 'voxels', coords.shape[0])
for k, v in T.items():
    print('%-10s %7.3f ms' % (k, float(np.median(v))))
