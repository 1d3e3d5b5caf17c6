"""The single-scan step over time: ms per step in groups of 10 (one synchronisation per group), device allocations and
reserved bytes per group -- is the 9.2 ms mode of `variants.single_scan` a state the process enters?  PRELUDE=5 runs a
5-scan training loop first (what bench.py does before the variant)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN, GeometryPrefetcher  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402

dev = torch.device('cuda', 0)
if os.environ.get('BIND'):
    from lidal_amd import backend as B
    print('bound to', len(B.bind_cpus_near(0) or ()), 'cpus')


def loop(frames, groups, tag):
    b = synth.make_train_batch(n_frames=frames, n_points=120000, seed=7122)
    coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
    torch.manual_seed(7122)
    model = SPVCNN(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    pf = GeometryPrefetcher(model, device=dev)
    g = pf.submit(coords)
    out = []
    for grp in range(groups):
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        a0 = st.get('num_device_alloc', 0)
        t0 = time.perf_counter()
        for _ in range(10):
            train_step(model, opt, feats, coords, labels, autocast=True, geometry=g)
            g = pf.submit(coords)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        st = torch.cuda.memory_stats(dev)
        out.append('%.2f' % dt + ('*%d' % (st.get('num_device_alloc', 0) - a0) if st.get('num_device_alloc', 0) != a0 else ''))
    pf.drain()
    print(tag, 'reserved %.1f GB |' % (torch.cuda.memory_reserved(dev) / 2**30), ' '.join(out), flush=True)


pre = int(os.environ.get('PRELUDE', '0'))
if pre:
    loop(pre, 3, 'prelude %d scans:' % pre)
loop(1, int(os.environ.get('GROUPS', '30')), 'one scan:')
loop(1, 10, 'one scan, a second model:')
