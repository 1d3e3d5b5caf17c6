"""Host round trips inside Geometry.build: which calls copy a value to the host (item / tolist / cpu / int(tensor))?"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidal_amd import synth
from lidal_amd.network import SPVCNN, MinkUNet, Geometry

dev = torch.device('cuda')
b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
log = []


def wrap(name):
    real = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if self.is_cuda:
            t0 = time.perf_counter()
            out = real(self, *a, **k)
            fr = [l for l in traceback.extract_stack()[:-1] if 'lidal_amd' in l.filename][-1]
            log.append((name, (time.perf_counter() - t0) * 1e3, '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name)))
            return out
        return real(self, *a, **k)
    setattr(torch.Tensor, name, f)


for n in ('item', 'tolist', 'cpu', '__int__', '__bool__', '__index__'):
    wrap(n)
for cls in (SPVCNN, MinkUNet):
    model = cls(19).to(dev).train()
    for rep in range(3):
        log.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Geometry.build(model, coords, grad=True)
        host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
    print(cls.__name__, 'build: host %.2f ms, %d host round trips:' % (host, len(log)))
    for name, ms, where in log:
        print('   %-8s %.3f ms  %s' % (name, ms, where))
