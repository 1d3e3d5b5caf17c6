"""Do N training steps on changing batches give the same losses bit for bit -- run to run, and with the coordinate
tables prefetched on the second stream?  Prints the first step at which two runs differ."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, GeometryPrefetcher
from lidal_amd.train_step import train_step

dev = torch.device('cuda')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
EVAL = len(sys.argv) > 2
batches = []
for i in range(6):
    b = synth.make_train_batch(n_frames=2, n_points=60000 + 7000 * i, seed=100 + i)
    batches.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
torch.manual_seed(0)
base = SPVCNN(19).to(dev).train()


def run(prefetch, fused_adam=True):
    model = copy.deepcopy(base)
    opt = torch.optim.Adam(model.parameters(), fused=fused_adam)
    torch.manual_seed(1)
    pf = GeometryPrefetcher(model) if prefetch else None
    g = pf.submit(batches[0][1]) if prefetch else None
    out = []
    for s in range(steps):
        f, c, lab = batches[s % len(batches)]
        loss, _ = train_step(model, opt, f, c, lab, autocast=True, geometry=g)
        if prefetch:
            g = pf.submit(batches[(s + 1) % len(batches)][1])
        out.append(loss)
        if EVAL and (s + 1) % 25 == 0:          # an evaluation pass in between (tables built in line), as scripts/soak.py
            import lidal_amd
            model.eval()
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                model(lidal_amd.SparseTensor(f, c))
            model.train()
    torch.cuda.synchronize()
    return [float(v) for v in out]


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return 'step %d: %.9g vs %.9g' % (i, x, y)
    return 'identical over %d steps' % len(a)


a, b = run(False), run(False)
print('inline vs inline    :', first_diff(a, b))
c, d = run(True), run(True)
print('prefetch vs prefetch:', first_diff(c, d))
print('inline vs prefetch  :', first_diff(a, c))
