"""Child process of tests/test_multirank_gpu.py: one rank of a 2-rank gloo group, both ranks on
cuda:0 (the N>1 code path on a 1-GPU box; on an 8-GPU node the same code runs with backend nccl
and one device per rank).  argv: mode out_dir.  Not collected by pytest (no test_ prefix)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import multirank_common as mc  # noqa: E402


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    if mode in ('score', 'score_allgather', 'score16'):
        from lidal_amd.score import collect_sequence, frame_range, score_sequence
        model = mc.make_model(dev, 16 if mode == 'score16' else 19).eval()    # (16 classes: the reference's nuScenes models)
        frames = mc.make_frames()
        mine = list(frame_range(len(frames), world, rank))
        local = [mc.to_device(frames[f], dev) for f in mine]
        first = mine[0] if mine else 0
        scores = score_sequence(model, local, first, len(frames), nei_num=mc.NEI, dis_thresh=0.1,
                                inf_reps=mc.REPS, autocast=False,
                                exchange='allgather' if mode == 'score_allgather' else 'halo')
        got = collect_sequence(scores, [frames[f]['sv_id'] for f in mine],
                               [d['sv_ptr'] for d in local], first, len(frames))
        if rank == 0:
            np.savez(os.path.join(out_dir, '%s_2rank.npz' % mode),
                     **{'%s_%d' % (k, f): v for f, t in enumerate(got)
                        for k, v in zip(('id', 'd', 'e', 'n', 'c'), t)})
    elif mode in ('ddp', 'dp', 'dp_per_operator', 'dp_mixed'):
        from lidal_amd.data_parallel import DataParallel
        from lidal_amd.network import plan
        from lidal_amd.train_step import forward_backward
        model = mc.make_model(dev).train()
        if rank == 1 and mode != 'ddp':         # (the wrapper must bring rank 0's parameters over, as DDP does)
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.5)
        model.dropout.p = 0.0
        if mode == 'ddp':
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
        else:
            net = DataParallel(model)
            # 'dp_mixed': rank 0 plans (in-place reduction of the flat buffer), rank 1 does not (reduction through a copy)
            plan.ENABLED = mode == 'dp' or (mode == 'dp_mixed' and rank == 0)
        b = mc.make_half_batches()[rank]
        net.zero_grad()
        loss, _ = forward_backward(net, b['feats'].to(dev), b['coords'].to(dev), b['labels'].to(dev))
        torch.cuda.synchronize()
        if mode != 'ddp':
            # the planned step's gradients are reduced in place as one tensor; the per-operator path's through a copy
            assert net.reductions == 1 and net.flat_reductions == (1 if plan.ENABLED else 0), (net.reductions, net.flat_reductions)
        if rank == 0:
            named = dict(model.named_parameters())
            np.savez(os.path.join(out_dir, '%s_2rank.npz' % mode), loss=loss.item(),
                     **{k.replace('.', '/'): named[k].grad.float().cpu().numpy() for k in mc.GRAD_KEYS})
    elif mode in ('eval', 'eval_empty'):
        # evaluate.py:95-124: every rank accumulates the confusion matrix of ITS batches, one
        # all-reduce sums them; 'eval_empty' gives rank 1 no batch at all
        from lidal_amd.evaluate import evaluate_batches
        model = mc.make_model(dev)
        batches = mc.make_val_batches()
        mine = batches[rank::world] if mode == 'eval' else (batches if rank == 0 else [])
        conf, ious, miou = evaluate_batches(model, [{k: v.to(dev) for k, v in b.items()} for b in mine])
        np.savez(os.path.join(out_dir, '%s_rank%d.npz' % (mode, rank)), conf=conf, miou=miou)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
