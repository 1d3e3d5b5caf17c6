"""GPU experiment: is one forward + backward of SPVCNN bit-reproducible?  Runs the same step REPS times from the same state
and reports, per parameter (in module order), how many repetitions differ from the first -- the logits first."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from lidal_amd import synth                     # noqa: E402
from lidal_amd.network import SPVCNN, MinkUNet   # noqa: E402
from lidal_amd.train_step import forward_backward     # noqa: E402

dev = torch.device('cuda')
REPS = int(os.environ.get('REPS', '6'))
autocast = os.environ.get('DTYPE', 'f32') == 'bf16'
b = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '2')), n_points=int(os.environ.get('POINTS', '60000')), seed=100)
f, c, lab = (torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b'))
for cls in (SPVCNN, MinkUNet):
    torch.manual_seed(0)
    base = cls(19).to(dev).train()
    outs = []
    for r in range(REPS):
        model = copy.deepcopy(base)
        torch.manual_seed(1)
        loss, logits = forward_backward(model, f, c, lab, autocast=autocast)
        torch.cuda.synchronize()
        outs.append((float(loss), logits.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}))
    l0, y0, g0 = outs[0]
    print(cls.__name__, 'autocast' if autocast else 'f32', 'voxels', c.shape[0], 'losses', sorted(set(o[0] for o in outs)),
          'logits differ in', sum(not torch.equal(o[1], y0) for o in outs[1:]), 'of', REPS - 1)
    bad = []
    for k in g0:
        nd = sum(not torch.equal(o[2][k], g0[k]) for o in outs[1:])
        if nd:
            md = max(float((o[2][k].double() - g0[k].double()).abs().max()) for o in outs[1:])
            bad.append((k, nd, md, float(g0[k].abs().max())))
    print('  parameters whose gradient differs:', len(bad), 'of', len(g0))
    for k, nd, md, sc in bad[-12:]:
        print('    %-46s differs in %d runs, max |d| %.3e (scale %.3e)' % (k, nd, md, sc))
