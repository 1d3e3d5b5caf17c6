"""Why is the host-bound single-scan step bimodal (6.7 vs 9.3 ms between processes on one box)?  Prints where the process
runs (CPU, NUMA node of the GPU, the GPU's local CPUs) and times the step; AFFINITY=local pins the process to the GPU's
local CPUs first, AFFINITY=<list> to a given cpu list."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def gpu_pci_dirs():
    out = []
    for d in sorted(glob.glob('/sys/class/drm/card*/device')):
        try:
            if open(os.path.join(d, 'vendor')).read().strip() == '0x1002':
                out.append(d)
        except OSError:
            pass
    return out


def read(path):
    try:
        return open(path).read().strip()
    except OSError as e:
        return 'n/a (%s)' % e.__class__.__name__


def parse_cpulist(s):
    cpus = set()
    for part in s.split(','):
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        elif part.strip():
            cpus.add(int(part))
    return cpus


dirs = gpu_pci_dirs()
info = [(d, read(d + '/numa_node'), read(d + '/local_cpulist')) for d in dirs]
aff = os.environ.get('AFFINITY')
if aff:
    cpus = parse_cpulist(info[0][2] if aff == 'local' else aff)
    os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
import torch  # noqa: E402

from lidal_amd import synth  # noqa: E402
from lidal_amd.network import SPVCNN, GeometryPrefetcher  # noqa: E402
from lidal_amd.train_step import train_step  # noqa: E402

dev = torch.device('cuda', 0)
b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
torch.manual_seed(7122)
model = SPVCNN(19).to(dev).train()
opt = torch.optim.Adam(model.parameters(), fused=True)
pf = GeometryPrefetcher(model, device=dev)
g = pf.submit(coords)
res = []
cpus_seen = set()
for rep in range(4):
    for _ in range(5):
        train_step(model, opt, feats, coords, labels, autocast=True, geometry=g)
        g = pf.submit(coords)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        train_step(model, opt, feats, coords, labels, autocast=True, geometry=g)
        g = pf.submit(coords)
        cpus_seen.add(int(open('/proc/self/stat').read().rsplit(')', 1)[1].split()[36]))
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 30 * 1e3)
print('ms/step %s | cpus seen %s | affinity %d cpus | gpu sysfs %s | nodes %s' % (
    ' '.join('%.2f' % r for r in res), sorted(cpus_seen), len(os.sched_getaffinity(0)), info[:2],
    read('/sys/devices/system/node/online')), flush=True)
