#!/usr/bin/env python
"""bench.py -- headline benchmark of the LiDAL sparse-voxel hot path on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one iteration of /root/reference/train.py:127-140 on a synthetic batch shaped like
the reference's (5 SemanticKITTI-shaped scans of ~120 k points, 0.05 m voxels, batch index as 4th
coordinate): zero_grad, SPVCNN forward (kernel maps rebuilt every step, as the reference's fresh
augmentation forces), cross-entropy(ignore 255), backward, Adam step -- conv operands bf16 with
f32 accumulation (BASELINE.json configs[1]).  value = input voxels of all ranks / second.

One JSON line is printed by rank 0.  Besides the contract fields it carries
  roofline      the dominant kernel (fused sparse conv, the level-0 96->96 k3 layer) timed live
                with HIP events on the launch stream: algorithmic bytes / duration
  cpu_baseline  the oracle (CPU restatement of the torchsparse path) on a bounded sample
  secondary     frames/s of prob_inference (8 views) + LiDAL inter-frame scoring, frame-sharded
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}


def log(*a):
    if os.environ.get('BENCH_VERBOSE'):
        print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--model', default='spvcnn', choices=['spvcnn', 'minkunet'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--frames', type=int, default=5, help='scans per step (sk_dataloader.py:21)')
    ap.add_argument('--points', type=int, default=120000)
    ap.add_argument('--score-frames', type=int, default=12, help='frames per rank for `secondary`')
    ap.add_argument('--nei', type=int, default=10, help='neighbour window (BASELINE config 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--roofline-only', action='store_true',
                    help='only the dominant-kernel measurement (for rocprofv3: every launch of the '
                         'kernel in the trace is then the roofline layer)')
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('BENCH_SINGLE_DEVICE'):       # test plumbing: N ranks on one GPU (gloo)
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or os.environ.get('BENCH_FORCE_DDP'):
        backend = os.environ.get('BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, 'launch with --nproc-per-node == --gpus (got %d vs %d)' % (world, args.gpus)
    return world, rank, torch.device('cuda', local)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()


def bench_train(args, world, rank, dev):
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet
    from lidal_amd.train_step import train_step
    batch = synth.make_train_batch(n_frames=args.frames, n_points=args.points, seed=7122 + rank)
    log('batch built', batch['coords_v_b'].shape)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    feats = torch.from_numpy(batch['feats_v_b']).to(dev)
    labels = torch.from_numpy(batch['labels_v_b']).to(dev)
    torch.manual_seed(7122)
    model = (SPVCNN if args.model == 'spvcnn' else MinkUNet)(19).to(dev).train()
    net = model
    if world > 1 or os.environ.get('BENCH_FORCE_DDP'):
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
    # Adam with the reference's defaults (train.py:56); `fused` only selects torch's single-kernel
    # implementation of the same update (the default path calls .item() once per parameter on the host)
    opt = torch.optim.Adam(net.parameters(), fused=True)
    autocast = args.dtype == 'bf16'

    def step():
        return train_step(net, opt, feats, coords, labels, autocast=autocast)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log('warmup step', i)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert np.isfinite(loss.item()), 'training diverged'
    voxels = sum_over_ranks(float(coords.shape[0]), world, dev)
    return {'model': model, 'coords': coords, 'seconds': dt, 'voxels_per_step': voxels,
            'loss': float(loss.item()), 'batch': batch}


def roofline_conv(args, coords, dev, reps=20):
    """Dominant kernel: the fused sparse conv (lidal_conv_apply) on the heaviest layer family --
    the 96->96 k3 convolutions at stride 1 (network/spvcnn.py:75-81).  Timed with HIP events on the
    stream the kernel is launched on (torch's current stream)."""
    from lidal_amd import backend as B
    from lidal_amd.nn import functional as F
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    b = 2 if args.dtype == 'bf16' else 4
    ci = co = 96
    kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    n, m = coords.shape[0], kmap.total
    order = kmap.order_out
    x = torch.randn(n, ci, device=dev).to(dtype)
    wk = (torch.randn(27, co, ci, device=dev) * 0.02).to(dtype)
    out = torch.empty((n, co), dtype=dtype, device=dev)

    def launch():
        B.check(B.lib().lidal_conv_apply(B.ptr(x), B.ptr(wk), B.ptr(order.table), B.ptr(order.perm),
                                         B.ptr(order.tile_masks), B.ptr(out), n, n, ci, co, 27, 0,
                                         B.dtype_code(dtype), None, None, 0, None, B.stream()), 'conv')
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / reps
    algo_bytes = b * (n * ci + n * co) + b * 27 * ci * co + 8 * m
    flops = 2.0 * m * ci * co
    gbs = algo_bytes / sec / 1e9
    traffic = None          # PMC-derived bytes per launch, measured offline on this exact workload
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_conv_apply.json')))
        wl = rec['workload']
        if (wl['rows'], wl['rules'], wl['dtype']) == (n, m, args.dtype):
            traffic = rec['traffic_bytes']
    except (OSError, KeyError, ValueError):
        pass
    return {'bound': 'hbm', 'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 5), 'traffic': traffic,
            'kernel': 'conv_apply_kernel (k3 s1 96->96, %s)' % args.dtype,
            'launch_us': round(sec * 1e6, 2), 'rows': n, 'rules': m,
            'algorithmic_bytes_per_launch': int(algo_bytes),
            'mfma': {'achieved': round(flops / sec / 1e12, 3), 'peak': MFMA_PEAK_TFLOPS[args.dtype],
                     'unit': 'TFLOP/s',
                     'frac': round(flops / sec / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5)}}


def cpu_baseline(args, rank_seed=7122, sample_points=None, max_threads=32):
    """Oracle (CPU restatement of the torchsparse path) on a BOUNDED sample of the same workload:
    one synthetic scan of `sample_points` points (same generator and input pipeline as the bench
    batch), forward + CE + backward once (upstream torchsparse has no CPU backward; autograd through
    the restatement supplies it).  Threads = min(host cores, max_threads)."""
    from lidal_amd import synth
    from oracle import tsref
    from oracle.models_ref import MinkUNetRef, SPVCNNRef
    cores = min(os.cpu_count() or 1, max_threads)
    sample_points = sample_points or args.points
    torch.set_num_threads(cores)
    batch = synth.make_train_batch(n_frames=1, n_points=sample_points, seed=rank_seed)
    coords = torch.from_numpy(batch['coords_v_b'])
    feats = torch.from_numpy(batch['feats_v_b'])
    labels = torch.from_numpy(batch['labels_v_b'])
    torch.manual_seed(7122)
    model = (SPVCNNRef if args.model == 'spvcnn' else MinkUNetRef)(19).train()
    t0 = time.perf_counter()
    logits, _ = model(tsref.SparseTensor(feats, coords))
    loss = torch.nn.functional.cross_entropy(logits, labels, ignore_index=255, reduction='mean')
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    n = coords.shape[0]
    return {'value': round(n / (t2 - t0), 1), 'unit': 'voxels/s', 'cores': cores, 'kind': 'port',
            'sample': '1 synthetic scan of %d points (%d voxels), %s f32 fwd+CE+bwd once on the '
                      'CPU oracle (fwd %.1f s, bwd %.1f s)' % (sample_points, n, args.model,
                                                               t1 - t0, t2 - t1)}


def bench_scoring(args, model, world, rank, dev):
    """prob_inference (8 augmented views per frame) + inter-frame scoring, frames sharded over
    ranks in the reference's contiguous blocks, probabilities/coords exchanged by one all-gather."""
    from lidal_amd import synth
    from lidal_amd.score import interframe, score_sequence
    per = args.score_frames
    total = per * world
    frames = synth.make_sequence(per, n_points=args.points, seed=7122, start=rank * per, total=total)
    rng = np.random.default_rng([7122, 99, rank])
    dev_frames = []
    for f in frames:
        sb = synth.make_score_batch(f['points'], f['intensity'], rng, inf_reps=8)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], dev)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(dev),
                           'feats': torch.from_numpy(sb['feats_v_b']).to(dev),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(dev),
                           'world': torch.from_numpy(f['world']).to(dev), 'sv_ptr': ptr, 'sv_idx': idx})
    model.eval()
    autocast = args.dtype == 'bf16'
    log('scoring inputs resident')

    def run():
        return score_sequence(model, dev_frames, rank * per, total, nei_num=args.nei, dis_thresh=0.1,
                              inf_reps=8, autocast=autocast)
    run()                                   # warm-up
    log('scoring warm-up done')
    barrier_sync(world)
    t0 = time.perf_counter()
    out = run()
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert all(torch.isfinite(o[0]).all() for o in out)
    model.train()
    return {'metric': 'frames/sec prob_inference(8 views)+LiDAL scoring', 'value': round(total / dt, 3),
            'unit': 'frames/s', 'frames': total, 'nei_num': args.nei, 'points_per_frame': args.points,
            'exchange': 'all_gather(prob f32 [P,19], world f64 [P,3])' if world > 1 else 'none (1 rank)'}


def main():
    args = parse()
    world, rank, dev = dist_setup(args)
    from lidal_amd import backend
    backend.lib()                           # fail loudly if the HIP library is missing
    if args.roofline_only:
        from lidal_amd import synth
        batch = synth.make_train_batch(n_frames=args.frames, n_points=args.points, seed=7122 + rank)
        coords = torch.from_numpy(batch['coords_v_b']).to(dev)
        print(json.dumps({'roofline': roofline_conv(args, coords, dev)}), flush=True)
        return
    res = bench_train(args, world, rank, dev)
    log('train timed: %.3f s for %d steps' % (res['seconds'], args.steps))
    ms = res['seconds'] / args.steps * 1e3
    line = {
        'metric': 'voxels/sec SPVCNN fwd+bwd @120k pts' if args.model == 'spvcnn'
                  else 'voxels/sec MinkUNet fwd+bwd @120k pts',
        'value': round(res['voxels_per_step'] * args.steps / res['seconds'], 1),
        'unit': 'voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s train step (train.py:127-140: fwd + CE + bwd + Adam), %d scans x ~%dk pts '
                               'per GPU, 0.05 m voxels, kernel maps rebuilt every step'
                               % (args.model, args.frames, args.points // 1000),
                   'voxels_per_step_per_gpu': int(res['voxels_per_step'] / world),
                   'parallelism': 'dp%d' % world, 'loss': round(res['loss'], 4)},
    }
    if rank == 0 and not args.no_roofline:
        line['roofline'] = roofline_conv(args, res['coords'], dev)
        log('roofline', line['roofline'])
    if not args.no_secondary:
        sec = bench_scoring(args, res['model'], world, rank, dev)
        log('secondary', sec)
        if rank == 0:
            line['secondary'] = sec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu baseline ...')
        line['cpu_baseline'] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
