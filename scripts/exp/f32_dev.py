"""GPU: which part of the f32 train step moves the gradients away from the f64 oracle at bench size?
One MinkUNet / SPVCNN step on a 120k-point scan under a few switches; prints per-parameter |g| deviations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_benchsize_gpu import GKEYS, _oracle_step  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'minkunet'
    from lidal_amd import backend as B
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.nn.functional import dense
    from lidal_amd.train_step import forward_backward
    from oracle.models_ref import MinkUNetRef, SPVCNNRef
    from weights import fill_state_dict
    b = synth.make_train_batch(n_frames=1, n_points=120000, seed=7122)
    coords, feats, labels = (torch.from_numpy(b[k]) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref_cls = {'spvcnn': SPVCNNRef, 'minkunet': MinkUNetRef}[name]
    loss64, logits64, g64 = _oracle_step(ref_cls, torch.float64, coords, feats, labels, GKEYS)
    loss32, logits32, g32 = _oracle_step(ref_cls, torch.float32, coords, feats, labels, GKEYS)

    def dev(gr):
        return {k: '%.1e' % abs(gr[k].double().cpu().norm().item() / g64[k].norm().item() - 1) for k in g64}
    print('cpu f32 oracle      ', dev(g32), 'logits %.1e' % ((logits32.double() - logits64).abs().max() / logits64.abs().max()))

    def run(tag):
        model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).cuda().train()
        if hasattr(model, 'dropout'):
            model.dropout.p = 0.0
        loss, logits = forward_backward(model, feats.cuda(), coords.cuda(), labels.cuda())
        named = dict(model.named_parameters())
        rel = ((logits.detach().double().cpu() - logits64).abs().max() / logits64.abs().max()).item()
        print('%-20s' % tag, dev({k: named[k].grad for k in g64}), 'logits %.1e loss %.2e' % (rel, abs(loss.item() / loss64 - 1)))
    run('hip f32')
    run('hip f32 again')
    saved = B._OVERLAP
    B._OVERLAP = '0'
    run('no side stream')
    B._OVERLAP = saved
    ok = dense._gemm_ok
    dense._gemm_ok = lambda x, ci, co: ok(x, ci, co) and x.dtype != torch.float32
    run('dense: library gemm')
    dense._gemm_ok = ok
    fork = B.FORK
    B.FORK = 0
    run('FORK=0')
    B.FORK = fork


if __name__ == '__main__':
    main()
