"""The N>1 code path executed on ONE MI355X: two fresh child processes form a gloo group on
cuda:0 (tests/_multirank_child.py) and must reproduce the single-process results.

  * scoring (configs 4/5): frames split in the reference's contiguous blocks
    (dataset/sk_dataloader.py:196-198), probabilities + world coordinates exchanged (halo exchange and
    all-gather), every rank
    scores its own frames, per-supervoxel results collected on rank 0 -- every number BIT-EQUAL to
    the 1-rank run (each frame's inference and scoring is the same kernel sequence on the same
    inputs whichever rank runs it; the kernels are order-deterministic);
  * training (configs 2/3): DistributedDataParallel as /root/reference/train.py:49-53 wraps the
    model (no SyncBatchNorm in the reference): the all-reduced gradient must equal the mean of the
    two ranks' single-process gradients within f32 rounding.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(mode, out_dir, world=2, timeout=900):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, '_multirank_child.py'), mode,
                                       str(out_dir)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, outs[r][-4000:])


def test_two_rank_scoring_is_bit_equal_to_one_rank(tmp_path):
    import multirank_common as mc
    from lidal_amd.score import collect_sequence, score_sequence
    dev = torch.device('cuda', 0)
    model = mc.make_model(dev).eval()
    frames = mc.make_frames()
    local = [mc.to_device(f, dev) for f in frames]
    scores = score_sequence(model, local, 0, len(frames), nei_num=mc.NEI, dis_thresh=0.1,
                            inf_reps=mc.REPS, autocast=False)
    one = collect_sequence(scores, [f['sv_id'] for f in frames], [d['sv_ptr'] for d in local], 0, len(frames))
    torch.cuda.synchronize()
    # both hand-offs: the halo exchange (default: each rank receives only the frames its block reads,
    # score/sharding.py HaloExchange) and the all-gather of every frame to every rank
    for mode in ('score', 'score_allgather'):
        _spawn(mode, tmp_path)
        two = np.load(os.path.join(str(tmp_path), '%s_2rank.npz' % mode))
        matched = 0
        for f, t in enumerate(one):
            for k, v in zip(('id', 'd', 'e', 'n', 'c'), t):
                assert np.array_equal(two['%s_%d' % (k, f)], v), (mode, k, f)
            matched += int((t[1] != 0).sum())
        assert matched > 0, 'degenerate fixture: no supervoxel saw an inter-frame match'


def test_two_rank_scoring_of_a_16_class_model(tmp_path):
    """The halo exchange sizes its receive buffers from the model's class count BEFORE any inference
    (score/pipeline.py _num_classes): a 16-class network (the reference's nuScenes configuration) must go through the
    2-rank path and reproduce the 1-rank scores bit for bit."""
    import multirank_common as mc
    from lidal_amd.score import collect_sequence, score_sequence
    dev = torch.device('cuda', 0)
    model = mc.make_model(dev, 16).eval()
    frames = mc.make_frames()
    local = [mc.to_device(f, dev) for f in frames]
    scores = score_sequence(model, local, 0, len(frames), nei_num=mc.NEI, dis_thresh=0.1, inf_reps=mc.REPS, autocast=False)
    one = collect_sequence(scores, [f['sv_id'] for f in frames], [d['sv_ptr'] for d in local], 0, len(frames))
    torch.cuda.synchronize()
    _spawn('score16', tmp_path)
    two = np.load(os.path.join(str(tmp_path), 'score16_2rank.npz'))
    for f, t in enumerate(one):
        for k, v in zip(('id', 'd', 'e', 'n', 'c'), t):
            assert np.array_equal(two['%s_%d' % (k, f)], v), (k, f)


@pytest.mark.parametrize('mode', ['ddp', 'dp', 'dp_per_operator', 'dp_mixed'])
def test_two_rank_ddp_gradient_is_the_mean_of_the_rank_gradients(tmp_path, mode):
    """'ddp': torch's DistributedDataParallel as train.py:49-53 wraps the model; 'dp': lidal_amd.data_parallel.DataParallel
    (one collective per backward pass on the planned step's flat gradient buffer); 'dp_per_operator': the same wrapper
    over the per-operator path (gradients reduced through a flattened copy); 'dp_mixed': rank 0 planned, rank 1 not -- the
    two forms of the reduction meet in ONE collective and must agree on its length and order."""
    import multirank_common as mc
    from lidal_amd.train_step import forward_backward
    dev = torch.device('cuda', 0)
    grads, losses = [], []
    for b in mc.make_half_batches():
        model = mc.make_model(dev).train()
        model.dropout.p = 0.0
        loss, _ = forward_backward(model, b['feats'].to(dev), b['coords'].to(dev), b['labels'].to(dev))
        named = dict(model.named_parameters())
        grads.append({k: named[k].grad.double().cpu().numpy() for k in mc.GRAD_KEYS})
        losses.append(loss.item())
    torch.cuda.synchronize()
    _spawn(mode, tmp_path)
    two = np.load(os.path.join(str(tmp_path), '%s_2rank.npz' % mode))
    assert abs(float(two['loss']) - losses[0]) <= 1e-6 * abs(losses[0])      # rank 0's own loss
    for k in mc.GRAD_KEYS:
        want = 0.5 * (grads[0][k] + grads[1][k])
        got = two[k.replace('.', '/')].astype(np.float64)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 1e-12, k


def test_two_rank_confusion_matrix_all_reduce(tmp_path):
    """SURVEY 8f-4's collective leg (evaluate.py:117-119): the 19x19 int32 confusion matrices of the
    ranks' own validation batches are summed by one all-reduce; every rank ends with the matrix (and
    mIoU) of the single-process run over all batches -- also when one rank has no batch."""
    import multirank_common as mc
    from lidal_amd.evaluate import evaluate_batches
    dev = torch.device('cuda', 0)
    model = mc.make_model(dev)
    one_conf, _, one_miou = evaluate_batches(model, [{k: v.to(dev) for k, v in b.items()}
                                                     for b in mc.make_val_batches()])
    assert one_conf.sum() > 0
    torch.cuda.synchronize()
    for mode in ('eval', 'eval_empty'):
        _spawn(mode, tmp_path)
        for r in range(2):
            got = np.load(os.path.join(str(tmp_path), '%s_rank%d.npz' % (mode, r)))
            assert np.array_equal(got['conf'], one_conf), (mode, r)
            assert (np.isnan(got['miou']) and np.isnan(one_miou)) or got['miou'] == one_miou
