"""GPU experiment: are several training steps (forward, backward, fused Adam; three batches in turn) bit-reproducible?
Reports the first step whose loss / parameters differ between repetitions, and which gradients differ at that step."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lidal_amd import synth                     # noqa: E402
from lidal_amd.network import SPVCNN             # noqa: E402
from lidal_amd.train_step import forward_backward     # noqa: E402

dev = torch.device('cuda')
REPS = int(os.environ.get('REPS', '5'))
STEPS = int(os.environ.get('STEPS', '6'))
autocast = os.environ.get('DTYPE', 'f32') == 'bf16'
batches = []
for i in range(3):
    b = synth.make_train_batch(n_frames=2, n_points=60000 + 7000 * i, seed=100 + i)
    batches.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
print('voxels', [int(b[1].shape[0]) for b in batches])
torch.manual_seed(0)
base = SPVCNN(19).to(dev).train()
if os.environ.get('NO_DROPOUT'):
    base.dropout.p = 0.0
runs = []
for r in range(REPS):
    model = copy.deepcopy(base)
    opt = torch.optim.Adam(model.parameters(), fused=not os.environ.get('NO_FUSED_ADAM'))
    torch.manual_seed(1)
    hist = []
    for s in range(STEPS):
        f, c, lab = batches[s % 3]
        opt.zero_grad()
        loss, logits = forward_backward(model, f, c, lab, autocast=autocast)
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        hist.append((float(loss.detach()), logits.detach().clone(), grads, {k: p.detach().clone() for k, p in model.named_parameters()}))
    runs.append(hist)
for s in range(STEPS):
    l0, y0, g0, p0 = runs[0][s]
    dl = sum(r[s][0] != l0 for r in runs[1:])
    dy = sum(not torch.equal(r[s][1], y0) for r in runs[1:])
    dg = [k for k in g0 if any(not torch.equal(r[s][2][k], g0[k]) for r in runs[1:])]
    dp = [k for k in p0 if any(not torch.equal(r[s][3][k], p0[k]) for r in runs[1:])]
    print('step %d: loss differs in %d runs, logits in %d, gradients of %d parameters, parameters after the update: %d'
          % (s, dl, dy, len(dg), len(dp)))
    if dg or dp:
        same = [k for k in g0 if k not in dg]
        print('   gradients that agree in every run (%d):' % len(same), ' '.join(same))
        which = [i for i, r in enumerate(runs[1:]) if any(not torch.equal(r[s][2][k], g0[k]) for k in g0)]
        print('   runs that differ from run 0:', which, 'of', REPS - 1)
        k = 'point_transforms.0.0.weight' if 'point_transforms.0.0.weight' in dg else dg[0]
        for r in runs[1:]:
            d = (r[s][2][k].double() - g0[k].double()).abs()
            print('   %s: max |d| %.3e at %s, scale %.3e, elements differing %d of %d'
                  % (k, float(d.max()), tuple(int(v) for v in (d == d.max()).nonzero()[0]), float(g0[k].abs().max()), int((d > 0).sum()), d.numel()))
        break
