"""Where does the time go when the next frame's coordinate tables are built on a second stream during inference?
Per-frame host time, GPU time, allocator statistics, in line and with the prefetcher.  (Round 3: with
record_stream on every table instead of the prefetcher's own hold-and-fence, 7.57 ms/frame against 7.23, and a pool
that kept growing; in line 8.15.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, GeometryPrefetcher
from lidal_amd.score.prob_inference import infer_frame

dev = torch.device('cuda:0')
model = SPVCNN(19).to(dev).eval()
N = int(os.environ.get('FRAMES', '8'))
frames = []
for f in synth.make_sequence(N, n_points=120000, seed=7122):
    sb = synth.make_score_batch(f['points'], f['intensity'], np.random.default_rng(1), inf_reps=8)
    frames.append(tuple(torch.from_numpy(sb[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'inverse_indices_b')))


def stats():
    s = torch.cuda.memory_stats()
    return 'reserved %.2f GB, device mallocs %d, frees %d' % (torch.cuda.memory_reserved() / 1e9, s['num_device_alloc'], s['num_device_free'])


def run(mode, reps=3):
    pf = None
    if mode != 'inline':
        pf = GeometryPrefetcher(model)
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = pf.submit(frames[0][0], grad=False) if pf else None
        host = []
        for i, (c, x, inv) in enumerate(frames):
            h0 = time.perf_counter()
            infer_frame(model, c, x, inv, 8, autocast=True, geometry=g)
            h1 = time.perf_counter()
            if pf:
                g = pf.submit(frames[(i + 1) % N][0], grad=False)
            h2 = time.perf_counter()
            host.append((h1 - h0, h2 - h1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%-26s pass %d: %.2f ms/frame   host: forward %.2f ms, submit %.2f ms   %s' % (
            mode, rep, dt / N * 1e3, np.mean([h[0] for h in host]) * 1e3, np.mean([h[1] for h in host]) * 1e3, stats()))


for mode in os.environ.get('MODES', 'inline,prefetch,inline').split(','):
    run(mode)
