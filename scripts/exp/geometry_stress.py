"""Are the coordinate tables built on the second stream, beside a busy main stream, bit for bit the tables built
alone?  Builds a reference geometry alone, then N geometries on the prefetcher's stream while training steps run on
the main stream, and compares every tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, Geometry, GeometryPrefetcher
from lidal_amd.train_step import train_step

dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = synth.make_train_batch(n_frames=2, n_points=67000, seed=101)
feats, coords, labels = (torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b'))
b2 = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
feats2, coords2, labels2 = (torch.from_numpy(b2[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b'))
torch.manual_seed(0)
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)


def tensors(obj, path, seen, out):
    if obj is None or isinstance(obj, (int, float, str, bool, torch.dtype, torch.device)) or id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        out.append((path, obj))
        for name in ('_lidal_invlist', '_lidal_i32'):
            tensors(getattr(obj, name, None), path + '.' + name, seen, out)
    elif isinstance(obj, dict):
        for k in sorted(obj, key=str):
            tensors(obj[k], '%s[%s]' % (path, k), seen, out)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            tensors(v, '%s[%d]' % (path, i), seen, out)
    elif hasattr(obj, '__dict__') and type(obj).__module__.startswith('lidal_amd'):
        for k in sorted(vars(obj)):
            tensors(vars(obj)[k], path + '.' + k, seen, out)


def table(g):
    out = []
    tensors({'x0': g.x0, 'z': g.z}, 'g', set(), out)
    return out


ref = Geometry.build(model, coords, grad=True)
torch.cuda.synchronize()
ref_t = [(p, t.clone()) for p, t in table(ref)]
again = Geometry.build(model, coords, grad=True)
torch.cuda.synchronize()
bad = [p for (p, a), (_, b_) in zip(ref_t, table(again)) if not torch.equal(a, b_)]
print('%d tensors per geometry; a second build alone differs in: %s' % (len(ref_t), bad or 'nothing'))
# keep the inputs of every trilinear-weight call so that it can be repeated alone afterwards
from lidal_amd.nn import functional as F
from lidal_amd.network import glue
_real = F.ti_weights_and_index
calls = []


MODE = int(os.environ.get('SPY_MODE', '0'))
stats = {'immediate_bad': 0, 'immediate_ok': 0}


def _spy(c, idx, scale=1):
    if MODE == 1:
        torch.cuda.current_stream().synchronize()       # everything the call reads has been produced
    w, i32 = _real(c, idx, scale)
    if MODE == 2:               # repeat at once, on the same stream, while the main stream is still busy
        w2, _ = _real(c, idx, scale)
        torch.cuda.current_stream().synchronize()
        stats['immediate_ok' if torch.equal(w, w2) else 'immediate_bad'] += 1
    calls.append((c, idx, scale, w, i32))
    return w, i32


glue.F.ti_weights_and_index = _spy
if os.environ.get('STREAM_PRIORITY') is not None:
    from lidal_amd.network import geometry as _geo
    _geo._STATE[torch.cuda.current_device()] = {
        'stream': torch.cuda.Stream(device=dev, priority=int(os.environ['STREAM_PRIORITY'])), 'orphans': []}
pf = GeometryPrefetcher(model)
print('second stream priority', pf.stream.priority)
g2 = pf.submit(coords2)
wrong = {}
for it in range(N):
    train_step(model, opt, feats2, coords2, labels2, autocast=True, geometry=g2)      # ~16 ms of main-stream work
    g = pf.submit(coords, grad=True)            # built beside it
    g2 = pf.submit(coords2)
    torch.cuda.synchronize()
    for c, idx, scale, w, i32 in calls:
        w2, i32b = _real(c, idx, scale)
        torch.cuda.synchronize()
        if not torch.equal(w, w2) or not torch.equal(i32, i32b):
            rows = torch.nonzero((w != w2).any(1)).flatten()
            r = int(rows[0]) if rows.numel() else 0
            print('  iteration %d scale %s n %d: same inputs repeated alone give other weights in %d rows (idx32 equal: %s; '
                  'idx32 == int64 idx: %s); row %d: beside %s  alone %s  idx64 %s' % (
                      it, scale, c.shape[0], rows.numel(), bool(torch.equal(i32, i32b)),
                      bool(torch.equal(i32, idx.t().int().contiguous())), r, w[r].tolist(), w2[r].tolist(), idx[:, r].tolist()))
    calls.clear()
    got = table(g)
    assert len(got) == len(ref_t)
    for (p, a), (q, b_) in zip(ref_t, got):
        if '_rules[0]' in p:
            continue                    # capacity arrays: only the first `total` rows are written
        if a.shape != b_.shape or not torch.equal(a, b_):
            n = int((a != b_).sum()) if a.shape == b_.shape else -1
            wrong.setdefault(p, []).append((it, n))
            if 'weights' in p and 'inv' not in p and len(wrong[p]) <= 2:
                rows = torch.nonzero((a != b_).any(1)).flatten()[:4].tolist()
                key = p[p.index('[(') + 1:p.index(')]') + 1]
                for r in rows:
                    print('   ', p, 'iteration', it, 'point', r, 'of', a.shape[0])
                    print('      alone ', a[r].tolist())
                    print('      beside', b_[r].tolist())
                    print('      coords', g.z.C[r].tolist(), ' ref coords', ref.z.C[r].tolist())
                    print('      idx   ', g.z.idx_query[eval(key)][r].tolist(), ' ref idx', ref.z.idx_query[eval(key)][r].tolist())
print('mode', MODE, stats)
print('iterations', N, ' tensors that differed from the build alone:', len(wrong))
for p, v in wrong.items():
    print('  ', p, v[:6])
