"""Every library call of one (in-line) train step with its integer arguments and event-timed duration, longest first
within a name filter.  usage: list_calls.py [substring ...]   e.g. list_calls.py voxel segment invlist"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidal_amd import backend as B, synth
from lidal_amd.network import SPVCNN
from lidal_amd.train_step import train_step

dev = 'cuda'
b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
torch.manual_seed(7122)
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)
for _ in range(3):
    train_step(model, opt, feats, coords, labels, autocast=True)
torch.cuda.synchronize()
calls = []


def val(a):
    v = getattr(a, 'value', a)
    try:
        return 0 if v is None else int(v)
    except (TypeError, ValueError):
        return -1


B.set_call_timer(lambda name, a, e0, e1: calls.append((name, [val(v) for v in a], e0, e1)))
train_step(model, opt, feats, coords, labels, autocast=True)
torch.cuda.synchronize()
B.set_call_timer(None)
want = sys.argv[1:]
rows = [(e0.elapsed_time(e1) * 1e3, n, [v for v in a if 0 <= v < 10 ** 7]) for n, a, e0, e1 in calls
        if not want or any(w in n for w in want)]
tot = sum(r[0] for r in rows)
print('%d calls, %.3f ms' % (len(rows), tot / 1e3))
for i, (us, n, a) in enumerate(rows):
    print('%4d %8.1f us  %-32s %s' % (i, us, n, a))
