"""Properties of the whole step that no single-operator test sees."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('name,autocast', [('spvcnn', True), ('minkunet', False)])
def test_results_do_not_depend_on_the_contents_of_fresh_buffers(name, autocast, monkeypatch):
    """Every buffer this package allocates comes from torch.empty / empty_like: with those filled with 0x00, 0xFF
    (NaN as floats, -1 as integers) or 0x7F bytes, three training steps must give the same losses and the same
    parameters bit for bit -- no kernel reads memory that neither it nor a kernel before it wrote (tile statistics of
    the last partial tile, capacity tails of the rule lists, padded weight images, workspaces of the sorts ...)."""
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.train_step import train_step
    pattern = [0]
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.numel() and t.is_contiguous():
            t.view(-1).view(torch.uint8).fill_(pattern[0])
        return t
    monkeypatch.setattr(torch, 'empty', lambda *a, **k: fill(real_empty(*a, **k)))
    monkeypatch.setattr(torch, 'empty_like', lambda *a, **k: fill(real_empty_like(*a, **k)))
    batches = []
    for i in range(2):
        b = synth.make_train_batch(n_frames=2, n_points=9000 + 1500 * i, seed=100 + i)
        batches.append(tuple(torch.from_numpy(b[k]).to(DEV) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
    torch.manual_seed(0)
    base = (SPVCNN if name == 'spvcnn' else MinkUNet)(19).to(DEV).train()

    def run(p):
        pattern[0] = p
        model = copy.deepcopy(base)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        torch.manual_seed(1)
        losses = []
        for s in range(3):
            f, c, lab = batches[s % 2]
            loss, _ = train_step(model, opt, f, c, lab, autocast=autocast)
            losses.append(float(loss))
        return losses, torch.cat([q.detach().flatten().float() for q in model.parameters()])
    (l0, p0), (l1, p1), (l2, p2) = run(0x00), run(0xFF), run(0x7F)
    assert l0 == l1 == l2, (l0, l1, l2)
    assert torch.equal(p0, p1) and torch.equal(p0, p2)


def test_cpu_binding_follows_the_gpu_topology():
    """backend.bind_cpus_near on the GPU box: the affinity of EVERY thread of the process becomes the GPU's local CPU list
    (sysfs), a subset of what the process was allowed before."""
    import os
    from lidal_amd import backend as B
    before = os.sched_getaffinity(0)
    try:
        cpus = B.bind_cpus_near(0)
        if cpus is None:
            pytest.skip('no local_cpulist for this GPU in sysfs')
        assert cpus and cpus <= before
        for tid in os.listdir('/proc/self/task'):
            assert os.sched_getaffinity(int(tid)) == cpus
    finally:
        for tid in os.listdir('/proc/self/task'):
            os.sched_setaffinity(int(tid), before)


def test_bench_prints_exactly_one_json_line_with_the_contract_keys():
    """bench.py's contract with the driver: ONE JSON line on stdout (whatever libraries print there on their own goes to
    stderr), carrying the metric, the roofline object and the extras.  A short run: 2 steps, no CPU baselines."""
    import json
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, BENCH_FORCE_DDP='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0')              # (one rank over RCCL: its version banner must not reach stdout)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', '--no-variants', '--no-families', '--no-secondary'],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['steps'] == 2 and d['n_gpus'] == 1 and d['value'] > 0 and d['config']['parallelism'] == 'dp1'
    r = d['roofline']
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic']
