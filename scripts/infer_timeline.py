"""Where does a prob_inference frame spend its wall time?  Splits one frame into its phases with
host timers + device syncs (coordinate pipeline, network, view mean)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lidal_amd import synth, SparseTensor
from lidal_amd.network import SPVCNN
from lidal_amd.score import prob_inference
dev = torch.device('cuda:0')
model = SPVCNN(19).to(dev).eval()
f = synth.make_sequence(1, n_points=120000, seed=7122, start=0, total=1)[0]
sb = synth.make_score_batch(f['points'], f['intensity'], np.random.default_rng(1), inf_reps=8)
c = torch.from_numpy(sb['coords_v_b']).to(dev); x = torch.from_numpy(sb['feats_v_b']).to(dev)
inv = torch.from_numpy(sb['inverse_indices_b']).to(dev)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for _ in range(3): prob_inference.infer_frame(model, c, x, inv, 8, autocast=True)
# whole frame, async vs fully synchronous launches
t0 = sync()
for _ in range(10): prob_inference.infer_frame(model, c, x, inv, 8, autocast=True)
t1 = sync(); print('frame, async: %.2f ms' % ((t1 - t0) / 10 * 1e3))
# host-only cost: how long until the python call returns (GPU may lag)
t0 = sync(); ts = []
for _ in range(10):
    a = time.perf_counter(); prob_inference.infer_frame(model, c, x, inv, 8, autocast=True); ts.append(time.perf_counter() - a)
t1 = sync(); print('python return per frame: %.2f ms (wall %.2f)' % (np.mean(ts) * 1e3, (t1 - t0) / 10 * 1e3))
# GPU span by events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
sync(); e0.record()
for _ in range(10): prob_inference.infer_frame(model, c, x, inv, 8, autocast=True)
e1.record(); sync(); print('GPU span by events: %.2f ms/frame' % (e0.elapsed_time(e1) / 10))
# the model alone
from lidal_amd.nn.functional.conv import prefetch_kernel_maps
with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
    for rep in range(3):
        t0 = sync()
        out = model(SparseTensor(x, c))
        t1 = sync()
    print('model forward alone (sync before/after): %.2f ms' % ((t1 - t0) * 1e3))
