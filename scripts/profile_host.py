"""GPU box: cProfile of the HOST side of a single-scan train step (the step is host-bound there:
~9 ms of kernels against 12-15 ms wall).  usage: profile_host.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda')
    b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
    coords = torch.from_numpy(b['coords_v_b']).to(dev)
    feats = torch.from_numpy(b['feats_v_b']).to(dev)
    labels = torch.from_numpy(b['labels_v_b']).to(dev)
    torch.manual_seed(7122)
    model = SPVCNN(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    for _ in range(5):
        train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    print('wall ms/step', (time.perf_counter() - t0) / steps * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime')
    print('---- by own time (ms per step) ----')
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:45]
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print('%8.3f ms  %7.1f calls  cum %8.3f  %s:%d %s' % (tt / steps * 1e3, nc / steps, ct / steps * 1e3,
                                                          fn[-40:], line, name))


if __name__ == '__main__':
    main()
