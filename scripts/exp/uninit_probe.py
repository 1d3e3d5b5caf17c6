"""Does any kernel read memory it (or a kernel before it) never wrote?  torch.empty / empty_like are patched to fill
every new buffer with a byte pattern; a training step whose result depends on the pattern has such a read.
usage: uninit_probe.py [module-substring ...]   (patch only allocations made from files whose path contains one)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

PATTERN = [0x00]
ONLY = sys.argv[1:]
_empty, _empty_like = torch.empty, torch.empty_like


def _want():
    if not ONLY:
        return True
    f = sys._getframe(2)
    for _ in range(6):
        if f is None:
            break
        if any(o in f.f_code.co_filename + ':' + f.f_code.co_name for o in ONLY):
            return True
        f = f.f_back
    return False


def _fill(t):
    if t.is_cuda and t.numel() and _want():
        t.view(-1).view(torch.uint8).fill_(PATTERN[0]) if t.is_contiguous() else None
    return t


torch.empty = lambda *a, **k: _fill(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_empty_like(*a, **k))

from lidal_amd import synth                     # noqa: E402
from lidal_amd.network import SPVCNN, MinkUNet   # noqa: E402
from lidal_amd.train_step import train_step     # noqa: E402
import copy                                     # noqa: E402

dev = torch.device('cuda')
batches = []
for i in range(3):
    b = synth.make_train_batch(n_frames=2, n_points=60000 + 7000 * i, seed=100 + i)
    batches.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
for cls in (SPVCNN, MinkUNet):
    torch.manual_seed(0)
    base = cls(19).to(dev).train()

    def run(pattern, autocast):
        PATTERN[0] = pattern
        model = copy.deepcopy(base)
        opt = torch.optim.Adam(model.parameters(), fused=True)
        torch.manual_seed(1)
        out = []
        for s in range(6):
            f, c, lab = batches[s % len(batches)]
            loss, logits = train_step(model, opt, f, c, lab, autocast=autocast)
            out.append(float(loss))
        torch.cuda.synchronize()
        return out, torch.cat([p.detach().flatten().float() for p in model.parameters()])

    for autocast in (True, False):
        (a, pa), (b, pb), (c, pc) = run(0x00, autocast), run(0xFF, autocast), run(0x7F, autocast)
        print(cls.__name__, 'bf16' if autocast else 'f32', 'losses 0x00 vs 0xFF:', a == b, ' vs 0x7F:', a == c,
              ' params equal:', bool(torch.equal(pa, pb)), bool(torch.equal(pa, pc)), a[:3], b[:3], c[:3])
