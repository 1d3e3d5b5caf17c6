"""The coordinate tables of a forward pass built AHEAD of it (lidal_amd/network/geometry.py), on the same or on a
second stream, against the in-line build inside `model(x)` (the reference's order of work: torchsparse builds its
maps on first use inside the forward pass, network/utils.py:13-102 the point tables): same results bit for bit,
for training steps on changing batches and for the 8-view inference step."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _models():
    from lidal_amd.network import SPVCNN, MinkUNet
    return {'spvcnn': SPVCNN, 'minkunet': MinkUNet}


def _batches(n, points=6000, frames=2):
    from lidal_amd import synth
    out = []
    for i in range(n):
        b = synth.make_train_batch(n_frames=frames, n_points=points + 700 * i, seed=100 + i)
        out.append(tuple(torch.from_numpy(b[k]).to(DEV) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
    return out


def _run(model, batches, autocast, prefetch):
    from lidal_amd.network import GeometryPrefetcher
    from lidal_amd.train_step import train_step
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    torch.manual_seed(5)                    # dropout masks (SPVCNN): the same in both runs
    losses, logits = [], None
    pf = GeometryPrefetcher(model) if prefetch else None
    g = pf.submit(batches[0][1]) if prefetch else None
    for i, (feats, coords, labels) in enumerate(batches):
        loss, logits = train_step(model, opt, feats, coords, labels, autocast=autocast, geometry=g)
        if prefetch and i + 1 < len(batches):
            g = pf.submit(batches[i + 1][1])        # beside the step just queued
        losses.append(loss)
    torch.cuda.synchronize()
    return [float(v) for v in losses], logits


@pytest.mark.parametrize('name,autocast', [('spvcnn', True), ('spvcnn', False), ('minkunet', True)])
def test_prefetched_geometry_train_steps_are_bitwise_the_inline_steps(name, autocast):
    torch.manual_seed(0)
    a = _models()[name](19).to(DEV).train()
    b = copy.deepcopy(a)
    batches = _batches(4)
    la, ya = _run(a, batches, autocast, prefetch=False)
    lb, yb = _run(b, batches, autocast, prefetch=True)
    assert la == lb, (la, lb)
    assert torch.equal(ya, yb)
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(p, q), k


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_prefetched_geometry_holds_every_table_the_forward_asks_for(name):
    """With a geometry the forward pass launches no map / table / list builder: counted at the C-ABI."""
    from lidal_amd import SparseTensor, backend as B
    from lidal_amd.network import Geometry
    torch.manual_seed(0)
    model = _models()[name](19).to(DEV).train()
    feats, coords, labels = _batches(1)[0]
    g = Geometry.build(model, coords)
    builders = ('lidal_kmap_build', 'lidal_kmap_build_batch', 'lidal_kmap_order', 'lidal_kmap_order_batch',
                'lidal_hash_table_build', 'lidal_hash_table_query', 'lidal_invlist_build', 'lidal_ti_weights',
                'lidal_downsample_pyramid', 'lidal_downsample', 'lidal_unique_sorted_i64', 'lidal_kmap_invert',
                'lidal_hash', 'lidal_kernel_hash', 'lidal_count', 'lidal_floor_coords', 'lidal_revoxelize_coords')
    assert all(b in B.SIGNATURES for b in builders)
    seen = []
    B.set_call_timer(lambda name, args, e0, e1: seen.append(name))
    try:
        x = SparseTensor(feats, coords)
        x.geometry = g
        logits, _ = model(x)
        logits.float().sum().backward()
    finally:
        B.set_call_timer(None)
    torch.cuda.synchronize()
    names = set(seen)
    assert names and not names & set(builders), sorted(names & set(builders))


def test_a_geometry_is_refused_for_other_coordinates_or_another_network():
    from lidal_amd import SparseTensor
    from lidal_amd.network import Geometry
    models = _models()
    torch.manual_seed(0)
    spv, mink = models['spvcnn'](19).to(DEV).eval(), models['minkunet'](19).to(DEV).eval()
    (f0, c0, _), (f1, c1, _) = _batches(2)
    g = Geometry.build(spv, c0)
    with torch.no_grad():
        x = SparseTensor(f1, c1)
        x.geometry = g
        with pytest.raises(RuntimeError, match='other coordinates'):
            spv(x)
        x = SparseTensor(f0, c0)
        x.geometry = g
        with pytest.raises(RuntimeError, match='built for SPVCNN'):
            mink(x)
        spv(x)


def test_prefetched_geometry_inference_step_is_bitwise():
    from lidal_amd import synth
    from lidal_amd.network import GeometryPrefetcher
    from lidal_amd.score.prob_inference import infer_frame
    torch.manual_seed(0)
    model = _models()['spvcnn'](19).to(DEV).eval()
    frames = []
    world = synth.make_world(5)
    for i in range(3):
        rng = np.random.default_rng(40 + i)
        pts, inten = synth.raycast_scan(world, (20.0 + 3 * i, 0.0), rng, n_beams=16 + 2 * i, n_az=128)
        d = synth.make_score_batch(pts, inten, rng, inf_reps=8)
        frames.append(tuple(torch.from_numpy(d[k]).to(DEV) for k in ('coords_v_b', 'feats_v_b', 'inverse_indices_b')))
    want = [infer_frame(model, c, f, inv, 8, autocast=True, return_feat=True) for c, f, inv in frames]
    pf = GeometryPrefetcher(model)
    g = pf.submit(frames[0][0])
    assert not g.grad
    for i, (c, f, inv) in enumerate(frames):
        got = infer_frame(model, c, f, inv, 8, autocast=True, return_feat=True, geometry=g)
        if i + 1 < len(frames):
            g = pf.submit(frames[i + 1][0])
        for a, b in zip(want[i], got):
            assert torch.equal(a, b)


def test_a_stale_prefetched_geometry_is_refused():
    """The prefetcher lets go of a geometry's memory after the second submit that follows its own: handing it to a
    forward pass after that must fail loudly, not read tables that another build may be writing."""
    from lidal_amd import SparseTensor
    from lidal_amd.network import GeometryPrefetcher
    torch.manual_seed(0)
    model = _models()['minkunet'](19).to(DEV).eval()
    (f0, c0, _), (f1, c1, _) = _batches(2)
    pf = GeometryPrefetcher(model)
    g0 = pf.submit(c0)
    g1 = pf.submit(c1)
    with torch.no_grad():
        x = SparseTensor(f0, c0)
        x.geometry = g0
        model(x)                        # one newer submission: fine
        pf.submit(c1)
        with pytest.raises(RuntimeError, match='stale'):
            model(x)
        x = SparseTensor(f1, c1)
        x.geometry = g1
        model(x)
        g2 = pf.submit(c1)
        pf.drain()                      # end of a loop: nothing is held any more, and what was is refused
        with pytest.raises(RuntimeError, match='stale'):
            x.geometry = g2
            model(x)
    torch.cuda.synchronize()


@pytest.mark.parametrize('planned', [True, False])
def test_a_backward_pass_over_a_stale_geometry_is_refused(planned):
    """The lifetime contract covers the BACKWARD pass too: two forwards, then two backwards (held graphs, micro-batch
    accumulation) queue the first backward after two later submits -- its tables may already serve another build.
    Refused with an error, on the planned step and on the per-operator path alike, not raced."""
    from lidal_amd import SparseTensor
    from lidal_amd.network import GeometryPrefetcher, plan
    from lidal_amd.nn.functional.fused import cross_entropy
    torch.manual_seed(0)
    model = _models()['minkunet'](19).to(DEV).train()
    (f0, c0, l0), (f1, c1, l1) = _batches(2)
    saved = plan.ENABLED
    plan.ENABLED = planned
    try:
        pf = GeometryPrefetcher(model)
        g0 = pf.submit(c0)
        x = SparseTensor(f0, c0)
        x.geometry = g0
        loss0 = cross_entropy(model(x)[0], l0)
        g1 = pf.submit(c1)
        x = SparseTensor(f1, c1)
        x.geometry = g1
        loss1 = cross_entropy(model(x)[0], l1)
        loss1.backward()                # one newer submission at most: fine
        pf.submit(c1)
        with pytest.raises(RuntimeError, match='stale'):
            loss0.backward()
        pf.drain()
    finally:
        plan.ENABLED = saved
    torch.cuda.synchronize()


def test_prefetchers_of_one_device_do_not_age_each_other():
    """Ages and fences are per prefetcher: a scorer's prefetcher run between a training loop's submit and its step must
    not make the loop's geometry stale (round 3 kept the held list per device)."""
    from lidal_amd import SparseTensor
    from lidal_amd.network import GeometryPrefetcher
    torch.manual_seed(0)
    model = _models()['minkunet'](19).to(DEV).eval()
    (f0, c0, _), (f1, c1, _) = _batches(2)
    a, b = GeometryPrefetcher(model), GeometryPrefetcher(model)
    g = a.submit(c0)
    for _ in range(3):
        b.submit(c1)
    with torch.no_grad():
        x = SparseTensor(f0, c0)
        x.geometry = g
        model(x)
    b.close()
    a.drain()
    del a, b
    torch.cuda.synchronize()


def _tables(g):
    """(path, tensor) of every table of a geometry; the rule lists up to their true length."""
    out, seen = [], set()

    def walk(obj, path):
        if obj is None or isinstance(obj, (int, float, str, bool, torch.dtype, torch.device)) or id(obj) in seen:
            return
        seen.add(id(obj))
        if isinstance(obj, torch.Tensor):
            out.append((path, obj))
            for name in ('_lidal_invlist', '_lidal_i32'):
                walk(getattr(obj, name, None), path + '.' + name)
        elif isinstance(obj, dict):
            for k in sorted(obj, key=str):
                walk(obj[k], '%s[%s]' % (path, k))
        elif isinstance(obj, (list, tuple)):
            for i, v in enumerate(obj):
                walk(v, '%s[%d]' % (path, i))
        elif type(obj).__name__ == 'KernelMap':
            for k in sorted(vars(obj)):
                if k == '_rules' and obj._rules is not None:
                    nbmaps, nbsizes, koff = obj._rules
                    walk(nbmaps[:int(koff[-1])], path + '._rules.nbmaps')       # capacity rows beyond `total`: never written
                    walk(nbsizes, path + '._rules.nbsizes')
                    walk(koff, path + '._rules.koff')
                else:
                    walk(vars(obj)[k], path + '.' + k)
        elif hasattr(obj, '__dict__') and type(obj).__module__.startswith('lidal_amd'):
            for k in sorted(vars(obj)):
                walk(vars(obj)[k], path + '.' + k)
    walk({'x0': g.x0, 'z': g.z}, 'g')
    return out


def test_tables_built_beside_the_bf16_convolutions_are_the_tables_built_alone():
    """The second stream runs beside the main stream's kernels.  On MI355X a wave executing v_mfma_f32_16x16x32_bf16
    disturbs packed-f32 instructions with op_sel of OTHER waves on its SIMD (profiles/README.md, round 3): hipcc had put
    one into ti_weights_kernel and ~1 % of the trilinear weights came out with a zero corner.  The library is built
    without packed f32 instructions; here every table of a geometry built beside a training step must equal, bit for
    bit, the table built on an idle GPU."""
    from lidal_amd.network import Geometry, GeometryPrefetcher
    from lidal_amd.train_step import train_step
    torch.manual_seed(0)
    model = _models()['spvcnn'](19).to(DEV).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    (f_big, c_big, l_big), = _batches(1, points=60000, frames=4)
    (_, c_small, _), = _batches(1, points=30000, frames=2)
    ref = Geometry.build(model, c_small, grad=True)
    torch.cuda.synchronize()
    want = [(p, t.clone()) for p, t in _tables(ref)]
    pf = GeometryPrefetcher(model)
    g_big = pf.submit(c_big)
    for it in range(25):
        train_step(model, opt, f_big, c_big, l_big, autocast=True, geometry=g_big)       # bf16 convolutions on the main stream
        g = pf.submit(c_small, grad=True)                                                   # ... and the tables beside them
        g_big = pf.submit(c_big)
        torch.cuda.synchronize()
        got = _tables(g)
        assert [p for p, _ in got] == [p for p, _ in want]
        for (p, a), (_, b) in zip(want, got):
            assert torch.equal(a, b), (it, p, int((a != b).sum()) if a.shape == b.shape else (a.shape, b.shape))
