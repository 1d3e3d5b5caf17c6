"""GPU: soak test -- many train steps on CHANGING batches with an evaluation every 50 steps; prints
step time and allocator state (a leak in the per-parameter / per-tensor caches would show here).
usage: soak.py [steps] [prefetch]   -- `prefetch`: every step's coordinate tables are built one step ahead on the
second stream (network/geometry.py); the losses must be the same numbers as without."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lidal_amd  # noqa: E402
from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN, GeometryPrefetcher  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    prefetch = len(sys.argv) > 2 and sys.argv[2] == 'prefetch'
    dev = torch.device('cuda')
    batches = []
    for i in range(6):
        b = synth.make_train_batch(n_frames=2, n_points=60000 + 7000 * i, seed=100 + i)
        batches.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
    torch.manual_seed(0)
    model = SPVCNN(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    pf = GeometryPrefetcher(model) if prefetch else None
    g = pf.submit(batches[0][1]) if prefetch else None
    t0 = time.perf_counter()
    for s in range(steps):
        f, c, lab = batches[s % len(batches)]
        loss, _ = train_step(model, opt, f, c, lab, autocast=True, geometry=g)
        if prefetch:
            g = pf.submit(batches[(s + 1) % len(batches)][1])
        if (s + 1) % 50 == 0:
            model.eval()
            with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                logits, _ = model(lidal_amd.SparseTensor(f, c))
            model.train()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 50 * 1e3
            t0 = time.perf_counter()
            print('step %4d  loss %.4f  %.2f ms/step  allocated %.0f MB  reserved %.0f MB  argmax-classes %d'
                  % (s + 1, loss.item(), dt, torch.cuda.memory_allocated() / 2 ** 20,
                     torch.cuda.memory_reserved() / 2 ** 20, logits.argmax(1).unique().numel()), flush=True)
    assert torch.isfinite(loss)


if __name__ == '__main__':
    main()
