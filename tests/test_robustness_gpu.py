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
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    # counter traffic: this round's offline record if it was taken on these kernel sources -- else null WITH the reason
    # (round 6: a record of an older build is never quoted, bench.py pmc_record)
    if r['traffic'] is None:
        assert 'profiles/r' in r['traffic_source'] and ('STALE' in r['traffic_source'] or 'no counter record' in r['traffic_source']
                                                        or 'another workload' in r['traffic_source'])
    else:
        assert r['traffic'] > r['algorithmic_bytes_per_launch'] and r['traffic_source'].startswith('offline PMC passes')


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE in the environment): the parent starts two fresh rank
    processes before touching the GPU (as train.py:163-203 / prob_inference.py:219-223 mp.spawn theirs), relays rank 0's
    ONE line and reports failure through its exit code.  Two gloo ranks on the one device of the test box."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(BENCH_SINGLE_DEVICE='1', BENCH_BACKEND='gloo', BENCH_LAUNCH_TIMEOUT='900')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline', '--score-frames', '7', '--nei', '10', '--points', '30000'],
                         capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['parallelism'] == 'dp2' and d['value'] > 0
    sec = d['secondary']
    assert sec['dtype'] == 'f32' and sec['frames'] == 14 and sec['value'] == sec['by_dtype']['f32']['by_nei']['10']['value']
    assert sec['by_dtype']['bf16']['by_nei']['10']['value'] > 0
    # a rank that dies takes the job down with a non-zero exit code and no line
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline', '--no-secondary', '--model', 'spvcnn', '--points', '30000'],
                         capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(env, BENCH_BACKEND='no-such-backend'))
    assert bad.returncode != 0 and not bad.stdout.strip()


def test_two_models_training_on_two_streams_in_one_process():
    """Two models trained step by step on two streams of one device, never synchronised against each other, for 200
    steps each: their fused BatchNorm launches (bn.hip: the merge inside the consumer, values published through
    buffers that belong to the launch's STREAM) overlap freely, hundreds of launch tokens apart.  Each model must end
    bit for bit where the same model ends when trained alone -- no overwritten publication buffer, no NaN, and the
    device's error word stays clear."""
    from lidal_amd import backend as B
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.train_step import train_step
    steps = 200
    batches = []
    for i in range(3):
        b = synth.make_train_batch(n_frames=2, n_points=5000 + 700 * i, seed=300 + i)
        batches.append(tuple(torch.from_numpy(b[k]).to(DEV) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
    torch.manual_seed(0)
    base = {'a': SPVCNN(19).to(DEV).train(), 'b': MinkUNet(19).to(DEV).train()}

    def make(name):
        model = copy.deepcopy(base[name])
        return model, torch.optim.Adam(model.parameters(), lr=1e-4)

    def flat(model):
        return torch.cat([q.detach().flatten().float() for q in model.parameters()] +
                         [q.detach().flatten().float() for q in model.buffers()])

    def one(model, opt, s, gen):
        f, c, lab = batches[s % 3]
        # (SPVCNN's dropout draws from the device generator: give every (model, step) its own seed in both runs)
        torch.manual_seed(gen * 100003 + s)
        return train_step(model, opt, f, c, lab, autocast=True)[0]
    alone = {}
    for gen, name in enumerate(('a', 'b')):
        model, opt = make(name)
        for s in range(steps):
            loss = one(model, opt, s, gen)
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
        alone[name] = flat(model)
    streams = {'a': torch.cuda.Stream(), 'b': torch.cuda.Stream()}
    pair = {name: make(name) for name in ('a', 'b')}
    torch.cuda.synchronize()
    for s in range(steps):
        for gen, name in enumerate(('a', 'b')):
            with torch.cuda.stream(streams[name]):
                one(pair[name][0], pair[name][1], s, gen)
    torch.cuda.synchronize()
    assert B.lib().lidal_bn_check_device() == 0, B.lib().lidal_last_error()
    for name in ('a', 'b'):
        got = flat(pair[name][0])
        assert torch.isfinite(got).all()
        assert torch.equal(got, alone[name]), name
