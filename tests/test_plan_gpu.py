"""The planned training step (lidal_amd/network/plan.py: the whole forward pass and the whole backward pass each
ONE lidal_plan_run call, one autograd node for the network) against the per-operator path it replaces (one
Python call per operator, as /root/reference/train.py:127-140 drives torchsparse): the plan issues the same entry
points with the same arguments, so loss, logits, every parameter gradient, every BatchNorm buffer and the
parameters after the optimizer steps must be BITWISE equal -- both networks, f32 and bf16, with the dropout of
SPVCNN active, on changing batches, with the coordinate tables built in line or ahead on a second stream."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _models():
    from lidal_amd.network import SPVCNN, MinkUNet
    return {'spvcnn': SPVCNN, 'minkunet': MinkUNet}


def _batches(n, points=9000, frames=2):
    from lidal_amd import synth
    out = []
    for i in range(n):
        b = synth.make_train_batch(n_frames=frames, n_points=points + 900 * i, seed=200 + i)
        out.append(tuple(torch.from_numpy(b[k]).to(DEV) for k in ('feats_v_b', 'coords_v_b', 'labels_v_b')))
    return out


def _nodes(loss):
    """(autograd nodes, custom Function nodes) reachable from the loss."""
    n = custom = 0
    seen, stack = set(), [loss.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        n += 1
        custom += hasattr(fn, '_forward_cls')          # a torch.autograd.Function node
        stack.extend(f for f, _ in fn.next_functions)
    return n, custom


def _steps(model, batches, autocast, planned, prefetch=False, steps_with_grads=()):
    """Train `len(batches)` steps; returns losses, last logits, gradients of the requested steps, library calls."""
    from lidal_amd import backend as B
    from lidal_amd.network import GeometryPrefetcher, plan
    from lidal_amd.train_step import forward_backward
    saved = plan.ENABLED, plan.TALLY
    plan.ENABLED = planned
    plan.TALLY = False                          # (HITS then counts the calls that really crossed from Python to the library)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        torch.manual_seed(11)                   # the dropout masks of SPVCNN: the same in both runs
        pf = GeometryPrefetcher(model) if prefetch else None
        g = pf.submit(batches[0][1]) if prefetch else None
        losses, grads, nodes, calls = [], {}, None, None
        for i, (feats, coords, labels) in enumerate(batches):
            opt.zero_grad()
            B.HITS.clear()
            loss, logits = forward_backward(model, feats, coords, labels, autocast=autocast, geometry=g)
            if prefetch and i + 1 < len(batches):
                g = pf.submit(batches[i + 1][1])
            if i == 0:
                nodes = _nodes(loss)
            calls = dict(B.HITS)                # of the last step (the first one also registers the weight images)
            if i in steps_with_grads:
                grads[i] = [p.grad.clone() for p in model.parameters()]
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return [float(v) for v in losses], logits.detach().clone(), grads, nodes, calls
    finally:
        plan.ENABLED, plan.TALLY = saved


@pytest.mark.parametrize('name,autocast', [('spvcnn', True), ('spvcnn', False), ('minkunet', True), ('minkunet', False)])
def test_planned_steps_are_bitwise_the_per_operator_steps(name, autocast):
    torch.manual_seed(0)
    a = _models()[name](19).to(DEV).train()
    b = copy.deepcopy(a)
    batches = _batches(3)
    la, ya, ga, na, ca = _steps(a, batches, autocast, planned=False, steps_with_grads=(0, 2))
    lb, yb, gb, nb, cb = _steps(b, batches, autocast, planned=True, steps_with_grads=(0, 2))
    assert cb.get('plan_run', 0) >= 2 and 'plan_run' not in ca, (ca, cb)        # the plan really ran / really did not
    assert la == lb, (la, lb)
    assert torch.equal(ya, yb)
    names = [k for k, _ in a.named_parameters()]
    for step in (0, 2):
        for k, p, q in zip(names, ga[step], gb[step]):
            assert torch.equal(p, q), (step, k, (p.float() - q.float()).abs().max().item())
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):          # parameters after 3 Adam steps, buffers
        assert torch.equal(p, q), k
    # one node for the network + the loss (the per-operator path: ~41 block nodes + glue)
    assert nb[1] <= 2 and na[1] >= 20, (na, nb)
    # library calls made from Python in one forward + backward pass (both build the coordinate tables in line here:
    # ~50 calls; test_planned_steps_with_prefetched_geometry counts the planned step without them)
    assert sum(ca.values()) >= 150, ca
    assert sum(cb.values()) <= 70, cb


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_planned_steps_with_prefetched_geometry(name):
    """The tables built one step ahead on the second stream (GeometryPrefetcher) under the planned step: the same
    numbers as the planned step that builds them in line."""
    torch.manual_seed(1)
    a = _models()[name](19).to(DEV).train()
    b = copy.deepcopy(a)
    batches = _batches(4, points=7000)
    la, ya, _, _, _ = _steps(a, batches, True, planned=True, prefetch=False)
    lb, yb, _, _, cb = _steps(b, batches, True, planned=True, prefetch=True)
    assert la == lb and torch.equal(ya, yb)
    for (k, p), q in zip(a.state_dict().items(), b.state_dict().values()):
        assert torch.equal(p, q), k
    # forward + backward plans (each cut in three around SPVCNN's dropouts), the weight images, the loss
    assert sum(cb.values()) <= (9 if name == 'spvcnn' else 5), cb
    assert cb['plan_run'] == (6 if name == 'spvcnn' else 2), cb


def test_planned_step_on_the_golden_fixture(golden_dir):
    """The planned step against the reference model files' float64 run (tests/golden/make_golden.py): the same bars as
    tests/test_model_gpu.py::test_train_step_matches_reference_golden holds the per-operator path to."""
    import os
    from lidal_amd import backend as B
    from lidal_amd.train_step import forward_backward
    from weights import fill_state_dict
    g = np.load(os.path.join(golden_dir, 'model_small.npz'))
    for name in ('spvcnn', 'minkunet'):
        model = fill_state_dict(_models()[name](19)).to(DEV).train()
        if hasattr(model, 'dropout'):
            model.dropout.p = 0.0
        B.HITS.clear()
        loss, logits = forward_backward(model, torch.from_numpy(g['feats']).to(DEV), torch.from_numpy(g['coords']).to(DEV),
                                        torch.from_numpy(g['labels']).to(DEV))
        assert B.HITS.get('plan_run', 0) == 2
        want = float(g[name + '_train_loss'])
        assert abs(loss.item() - want) < 1e-4 * abs(want)
        ref = g[name + '_train_logits'].astype(np.float64)
        got = logits.detach().cpu().numpy().astype(np.float64)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-4


def test_planned_step_refuses_a_second_backward_and_falls_back_when_not_plannable():
    from lidal_amd import SparseTensor
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.fused import cross_entropy
    feats, coords, labels = _batches(1, points=5000, frames=1)[0]
    model = _models()['minkunet'](19).to(DEV).train()
    logits, _ = model(SparseTensor(feats, coords))
    loss = cross_entropy(logits, labels)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='already run its backward'):
        loss.backward()
    # a frozen parameter, an input that wants a gradient: the per-operator path takes over (same module, no error)
    model.zero_grad()
    model.stem[0].kernel.requires_grad_(False)
    B.HITS.clear()
    logits, _ = model(SparseTensor(feats, coords))
    cross_entropy(logits, labels).backward()
    assert 'plan_run' not in B.HITS and model.stem[0].kernel.grad is None
    assert model.stem[3].kernel.grad is not None


def test_feature_output_gradient_flows_through_the_plan():
    """model(x) returns (logits, features); a loss on BOTH must reach the parameters through the planned node as
    through the per-operator graph."""
    from lidal_amd import SparseTensor
    from lidal_amd.network import plan
    from lidal_amd.nn.functional.fused import cross_entropy
    feats, coords, labels = _batches(1, points=6000, frames=1)[0]
    torch.manual_seed(3)
    a = _models()['spvcnn'](19).to(DEV).train()
    a.dropout.p = 0.0
    b = copy.deepcopy(a)
    out = []
    saved = plan.ENABLED
    try:
        for model, planned in ((a, False), (b, True)):
            plan.ENABLED = planned
            with torch.autocast('cuda', dtype=torch.bfloat16):
                logits, feat = model(SparseTensor(feats, coords))
            (cross_entropy(logits, labels) + feat.float().square().mean()).backward()
            out.append([p.grad.clone() for p in model.parameters()])
    finally:
        plan.ENABLED = saved
    for k, p, q in zip([k for k, _ in a.named_parameters()], *out):
        assert torch.equal(p, q), k


@pytest.mark.parametrize('name,autocast', [('spvcnn', True), ('spvcnn', False), ('minkunet', True)])
def test_planned_inference_is_bitwise_the_per_operator_inference(name, autocast):
    """The 8-view inference step (score/prob_inference.py:91-113) as ONE plan against the per-operator eval path
    (blocks.ConvNormSequential: folded BatchNorm epilogues): probabilities, predictions and point features bitwise,
    with the tables built in line and ahead, before and after the weights change."""
    from lidal_amd import backend as B
    from lidal_amd import synth
    from lidal_amd.network import GeometryPrefetcher, plan
    from lidal_amd.score.prob_inference import infer_frame
    torch.manual_seed(4)
    model = _models()[name](19).to(DEV).eval()
    for m in model.modules():               # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
    frames = []
    for i in range(2):
        seq = synth.make_sequence(1, n_points=9000 + 2000 * i, seed=30 + i)[0]
        sb = synth.make_score_batch(seq['points'], seq['intensity'], np.random.default_rng(i), inf_reps=8)
        frames.append(tuple(torch.from_numpy(sb[k]).to(DEV) for k in ('coords_v_b', 'feats_v_b', 'inverse_indices_b')))
    frames.append(frames[0])                # (sizes seen before: the last frame registers no new weight image)
    saved = plan.ENABLED, plan.TALLY
    plan.TALLY = False
    try:
        for round_ in range(2):
            outs = {}
            for planned in (False, True):
                plan.ENABLED = planned
                pf = GeometryPrefetcher(model)
                res = []
                for i, (c, f, inv) in enumerate(frames):
                    g = pf.submit(c, grad=False) if i >= 1 else None
                    B.HITS.clear()
                    res.append(infer_frame(model, c, f, inv, 8, autocast=autocast, return_feat=True, geometry=g))
                    calls = dict(B.HITS)
                pf.drain()
                outs[planned] = res
                assert (calls.get('plan_run', 0) == 1) == planned, calls
                if planned:
                    assert sum(calls.values()) <= 3, calls         # the plan, the view mean (+ the images when the weights moved)
            for a, b in zip(outs[False], outs[True]):
                for u, v in zip(a, b):
                    assert torch.equal(u, v)
            with torch.no_grad():               # the weights move: every cached operand must follow
                for p in model.parameters():
                    p.mul_(1.01)
    finally:
        plan.ENABLED, plan.TALLY = saved


def test_planned_step_with_every_shortcut_on_its_side_stream_is_bitwise():
    """plan.BRANCH_ROWS = 1: the shortcut branch of EVERY residual block (1x1x1 convolution + BatchNorm, forward and
    backward) on the third stream beside the main branch -- at the bench batch only the levels above 100 k rows go there.
    BatchNorm launches of two streams then run at the same time, each with its own publication slots (bn.hip): loss,
    logits and every gradient stay bitwise the per-operator path's."""
    from lidal_amd.network import plan
    torch.manual_seed(5)
    a = _models()['spvcnn'](19).to(DEV).train()
    b = copy.deepcopy(a)
    batches = _batches(3, points=12000, frames=2)
    saved = plan.BRANCH_ROWS
    try:
        plan.BRANCH_ROWS = 1
        la, ya, ga, _, _ = _steps(a, batches, True, planned=False, steps_with_grads=(2,))
        lb, yb, gb, _, cb = _steps(b, batches, True, planned=True, steps_with_grads=(2,))
    finally:
        plan.BRANCH_ROWS = saved
    assert cb.get('plan_run', 0) >= 2
    assert la == lb and torch.equal(ya, yb)
    for k, p, q in zip([k for k, _ in a.named_parameters()], ga[2], gb[2]):
        assert torch.equal(p, q), k


@pytest.mark.parametrize('train', [False, True])
def test_planned_forward_follows_a_change_of_one_weight_only(train):
    """A partial load_state_dict / a copy_ into ONE layer between two forward passes with no backward pass in between
    moves that layer's version counter only: the plan has to notice it weight by weight (the per-operator path's
    _ImageBank.get does), not on the first convolution alone -- stale LDS images would give wrong logits silently."""
    from lidal_amd import SparseTensor
    from lidal_amd.network import plan
    torch.manual_seed(6)
    model = _models()['spvcnn'](19).to(DEV)
    model.train(train)
    feats, coords, _ = _batches(1)[0]
    targets = ['stage3.1.net.3.kernel', 'up2.0.net.0.kernel', 'classifier.0.weight', 'point_transforms.1.0.weight']
    params = dict(model.named_parameters())
    assert all(t in params for t in targets), [k for k in params][:400]

    def forward(planned):
        saved = plan.ENABLED
        plan.ENABLED = planned
        try:
            torch.manual_seed(3)                # SPVCNN's dropout masks
            with torch.no_grad() if not train else torch.enable_grad():
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    out = model(SparseTensor(feats, coords))
            return (out[0] if isinstance(out, tuple) else out).detach().float().clone()
        finally:
            plan.ENABLED = saved
    first = forward(True)
    for t in targets:
        with torch.no_grad():
            params[t].mul_(1.5)
        got = forward(True)
        want = forward(False)
        assert not torch.equal(got, first), t
        assert torch.equal(got, want), t
        first = got
