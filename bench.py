#!/usr/bin/env python
"""bench.py -- headline benchmark of the LiDAL sparse-voxel hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
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
  families      one more (untimed) step with every library call bracketed by events: GPU time,
                algorithmic bytes / FLOPs and roofline fraction per kernel family
                (sparse conv fwd+dgrad, weight gradient, BatchNorm, kernel maps, point<->voxel,
                fused elementwise), and of the whole step
  variants      the same step on ONE scan (BASELINE.json's literal "@120k pts"), with a freshly
                augmented batch every step (new coordinates / sizes / kernel maps, as the reference's
                loader produces), in the f32 parity mode, and with the second backbone (MinkUNet, configs[2])
  secondary     frames/s of prob_inference (8 views) + LiDAL inter-frame scoring (configs[3..4]),
                frame-sharded, 32 frames per rank, neighbour windows 10 and 24, with its own CPU
                baseline (oracle worker_func restatement under a process pool, LiDAL.py:204-206)
  cpu_baseline  the oracle (CPU restatement of the torchsparse path) on a bounded sample
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
# the split form of an f32 product (LIDAL_F32_SPLIT): six bf16 MFMAs per f32 multiply-add block
SPLIT_PEAK_TFLOPS = 2500.0 / 6
ROUND = 6                      # offline counter records (profiles/rNN_pmc_*.json) count for their own round only


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
    ap.add_argument('--score-frames', type=int, default=32,
                    help='frames per rank for `secondary` (config 4: 256 frames / 8 GPUs)')
    ap.add_argument('--nei', type=int, nargs='+', default=[10, 24],
                    help='neighbour windows of `secondary` (BASELINE config 5: 10; LiDAL.py:120: 24)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--sequence-frames', type=int, default=256,
                    help='length of the one-GPU whole-sequence leg `variants.score_N` (config 4: 256; 0 = skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-families', action='store_true')
    ap.add_argument('--no-variants', action='store_true')
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


def collective_tensor(x, dev):
    cdev = dev if dist.get_backend() == 'nccl' else torch.device('cpu')
    return torch.tensor([x], dtype=torch.float64, device=cdev)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = collective_tensor(x, dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = collective_tensor(x, dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()


# ---------------------------------------------------------------------------------------------
# the train step
# ---------------------------------------------------------------------------------------------
def make_batch(frames, points, seed, dev):
    from lidal_amd import synth
    batch = synth.make_train_batch(n_frames=frames, n_points=points, seed=seed)
    return (torch.from_numpy(batch['coords_v_b']).to(dev), torch.from_numpy(batch['feats_v_b']).to(dev),
            torch.from_numpy(batch['labels_v_b']).to(dev))


def make_fresh_batches(frames, points, seed, dev, n_batches):
    """`n_batches` batches of the SAME `frames` scans, each under a newly drawn augmentation
    (dataset/sk_dataset.py:143-171: affine + flip + rotation, x20, random translation, int cast,
    unique rows), voxelised and collated ON THE GPU (lidal_voxelize_points, lidal_amd/data.py): new
    voxel coordinates, new row counts and new kernel maps every step, as in the reference's loop."""
    from lidal_amd import data, synth
    rng = np.random.default_rng(seed)
    world = synth.make_world(seed)
    scans = []
    for f in range(frames):
        pts, inten = synth.raycast_scan(world, (10.0 + 7.0 * f, 0.0), rng, n_points=points)
        labels_p = rng.integers(0, 19, size=pts.shape[0]).astype(np.int64)
        labels_p[rng.random(pts.shape[0]) < 0.1] = 255
        scans.append((torch.from_numpy(pts).to(dev), torch.from_numpy(inten).to(dev),
                      torch.from_numpy(labels_p).to(dev)))
    aug = np.random.RandomState(seed)
    out = []
    for _ in range(n_batches):
        samples = []
        for pts, inten, labels_p in scans:
            trans_m, rnd = data.draw_augmentation(aug)
            coords_v, feats_v, uniq, _ = data.voxelize_scan(pts, inten, trans_m, rnd)
            samples.append({'coords_v': coords_v, 'feats_v': feats_v, 'labels_v': labels_p[uniq]})
        b = data.collate(samples)
        out.append((b['coords_v_b'].contiguous(), b['feats_v_b'].contiguous(), b['labels_v_b'].contiguous()))
    torch.cuda.synchronize()
    return out


def bench_train(world, rank, dev, model_name, dtype, batch, steps, warmup, ddp=True, prefetch=True):
    """Times `steps` iterations of train.py:127-140 on the resident `batch` (coords, feats, labels) --
    or, if `batch` is a list of such tuples, on a different one of them every step.
    prefetch: every step builds the coordinate tables (voxel index, kernel maps, row orders, point <-> voxel
    tables) of the NEXT step's batch on a second stream right after queueing its own forward + backward
    (lidal_amd.network.GeometryPrefetcher) -- one build per step, as with prefetch=False, where the forward pass
    builds its own tables in line (the reference's order of work)."""
    from lidal_amd.network import SPVCNN, MinkUNet, GeometryPrefetcher
    from lidal_amd.train_step import train_step
    batches = batch if isinstance(batch, list) else [batch]
    torch.manual_seed(7122)
    model = (SPVCNN if model_name == 'spvcnn' else MinkUNet)(19).to(dev).train()
    net = model
    if ddp and (world > 1 or os.environ.get('BENCH_FORCE_DDP')):
        # train.py:49-53: gradients averaged over the ranks inside backward().  lidal_amd.data_parallel.DataParallel does
        # it with ONE collective on the planned step's flat gradient buffer; BENCH_TORCH_DDP=1: torch's wrapper
        if os.environ.get('BENCH_TORCH_DDP'):
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
        else:
            if world == 1:                      # BENCH_FORCE_DDP on one GPU: execute the collective over RCCL all the same
                os.environ.setdefault('LIDAL_DP_FORCE_COLLECTIVE', '1')
            from lidal_amd.data_parallel import DataParallel
            net = DataParallel(model)
    # Adam with the reference's defaults (train.py:56); `fused` only selects torch's single-kernel
    # implementation of the same update (the default path calls .item() once per parameter on the host)
    opt = torch.optim.Adam(net.parameters(), fused=True)
    autocast = dtype == 'bf16'
    count = [0]
    pf = GeometryPrefetcher(model, device=dev) if prefetch else None
    ahead = [pf.submit(batches[0][0]) if prefetch else None]

    delay = float(os.environ.get('BENCH_HOST_DELAY_US', '0')) * 1e-6     # experiment: is the step host-bound?  (a busy wait per step)

    def step():
        coords, feats, labels = batches[count[0] % len(batches)]
        count[0] += 1
        if delay:
            t_end = time.perf_counter() + delay
            while time.perf_counter() < t_end:
                pass
        out = train_step(net, opt, feats, coords, labels, autocast=autocast, geometry=ahead[0])
        if prefetch:
            ahead[0] = pf.submit(batches[count[0] % len(batches)][0])
        return out

    for i in range(warmup):
        step()
        torch.cuda.synchronize()
    barrier_sync(world if ddp else 1)
    first = count[0]
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step()
    t_host = time.perf_counter() - t0           # the host has queued every step (it runs ahead of the GPU unless it is the bound)
    barrier_sync(world if ddp else 1)
    dt = time.perf_counter() - t0
    if ddp:
        dt = max_over_ranks(dt, world, dev)
    assert np.isfinite(loss.item()), 'training diverged'
    voxels = sum(int(batches[(first + i) % len(batches)][0].shape[0]) for i in range(steps)) / steps
    return {'model': model, 'step': step, 'seconds': dt, 'steps': steps, 'voxels': voxels,
            'loss': float(loss.item()), 'host_seconds': t_host}


def host_calls(model_name, dtype, batch, dev, prefetch=True):
    """What one step costs the host in calls: autograd Function nodes behind the loss, library calls made from Python
    (backend.HITS: every ctypes call into liblidal_amd.so, the table build of the next step included) and the
    operations executed inside launch plans -- counted on one step after a warm-up one."""
    from lidal_amd import backend as B
    from lidal_amd.network import SPVCNN, MinkUNet, GeometryPrefetcher, plan
    from lidal_amd.train_step import forward_backward
    torch.manual_seed(7122)
    model = (SPVCNN if model_name == 'spvcnn' else MinkUNet)(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    coords, feats, labels = batch
    pf = GeometryPrefetcher(model, device=dev) if prefetch else None
    g = pf.submit(coords) if prefetch else None
    out = {}
    for it in range(2):
        opt.zero_grad()
        B.HITS.clear()
        ops0 = plan.COUNTERS['ops']
        loss, _ = forward_backward(model, feats, coords, labels, autocast=dtype == 'bf16', geometry=g)
        nodes, seen, stack = 0, set(), [loss.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            nodes += hasattr(fn, '_forward_cls')
            stack.extend(f for f, _ in fn.next_functions)
        opt.step()
        fb = sum(B.HITS.values())
        if prefetch:
            g = pf.submit(coords)
        out = {'autograd_function_nodes': int(nodes), 'library_calls_from_python': int(sum(B.HITS.values())),
               'of_which_forward_backward': int(fb), 'of_which_plan_run': int(B.HITS.get('plan_run', 0)),
               'operations_inside_plans': int(plan.COUNTERS['ops'] - ops0), 'launch_plan': bool(plan.ENABLED)}
    if pf is not None:
        pf.drain()
    torch.cuda.synchronize()
    return out


def bench_fresh_stream(dev, model_name, dtype, frames, points, steps, warmup=3, seed=7122):
    """The reference's real input stream inside the timed loop: EVERY step trains on a batch that did not exist before
    -- a new augmentation drawn on the host per scan (dataset/sk_dataset.py:143-147,156), the scans voxelised and
    collated on the GPU (lidal_voxelize_points, data.collate: sk_dataset.py:148-171,188-242) and the batch's coordinate
    tables built, all on the second stream beside the step before (GeometryPrefetcher.submit_batch) -- what the
    reference's DataLoader workers do ahead of the GPU (dataset/sk_dataloader.py:21,53).  No batch repeats; nothing is
    pre-warmed beyond `warmup` steps.  Reports ms/step, the allocator's reserved bytes and device allocations."""
    from lidal_amd import data, synth
    from lidal_amd.network import SPVCNN, MinkUNet, GeometryPrefetcher
    from lidal_amd.train_step import train_step
    rng = np.random.default_rng(seed)
    world = synth.make_world(seed)
    scans = []
    for f in range(frames):
        pts, inten = synth.raycast_scan(world, (10.0 + 7.0 * f, 0.0), rng, n_points=points)
        labels_p = rng.integers(0, 19, size=pts.shape[0]).astype(np.int64)
        labels_p[rng.random(pts.shape[0]) < 0.1] = 255
        scans.append((torch.from_numpy(pts).to(dev), torch.from_numpy(inten).to(dev), torch.from_numpy(labels_p).to(dev)))
    torch.manual_seed(7122)
    model = (SPVCNN if model_name == 'spvcnn' else MinkUNet)(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    aug = np.random.RandomState(seed)
    pf = GeometryPrefetcher(model, device=dev)

    def make():
        samples = []
        for pts, inten, labels_p in scans:
            trans_m, rnd = data.draw_augmentation(aug)
            coords_v, feats_v, uniq, _ = data.voxelize_scan(pts, inten, trans_m, rnd)
            samples.append({'coords_v': coords_v, 'feats_v': feats_v, 'labels_v': labels_p[uniq]})
        b = data.collate(samples)
        return {k: (v.contiguous() if v is not None else None) for k, v in b.items()}

    autocast = dtype == 'bf16'
    g = pf.submit_batch(make)
    sizes, reserved, segs = [], [], []

    def step():
        nonlocal g
        b = g.payload
        sizes.append(int(b['coords_v_b'].shape[0]))
        out = train_step(model, opt, b['feats_v_b'], b['coords_v_b'], b['labels_v_b'], autocast=autocast, geometry=g)
        g = pf.submit_batch(make)
        reserved.append(int(torch.cuda.memory_reserved(dev)))
        segs.append(int(torch.cuda.memory_stats(dev).get('segment.all.allocated', 0)))
        return out

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    first = len(sizes)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pf.drain()
    assert np.isfinite(loss.item())
    timed = sizes[first:]
    ms = dt / steps * 1e3
    return {'ms_per_step': round(ms, 3), 'steps': steps, 'warmup': warmup,
            'voxels_per_step': int(np.mean(timed)), 'voxels_per_s': round(float(np.sum(timed)) / dt, 1),
            'distinct_batches': len(set(sizes)), 'voxels_min_max': [int(min(timed)), int(max(timed))],
            'reserved_MB': {'after_warmup': reserved[first - 1] >> 20, 'step_10': reserved[min(first + 9, len(reserved) - 1)] >> 20,
                            'end': reserved[-1] >> 20},
            'device_allocations': {'after_warmup': segs[first - 1], 'step_10': segs[min(first + 9, len(segs) - 1)],
                                   'end': segs[-1]},
            'loss': round(float(loss.item()), 4),
            'what': 'every step: new augmentation per scan (host draws), GPU voxelisation + collate + coordinate tables of '
                    'the NEXT batch on the second stream; no batch repeats (sk_dataset.py:143-171, sk_dataloader.py:21,53)'}


def bench_dropin_surface(dev, model_name, dtype, batch, steps, warmup=3, adopt=False):
    """The literal drop-in path: a LiDAL user's network as the reference's files compose it (scripts/surface_unet.py:
    nn.Sequential(spnn.Conv3d, spnn.BatchNorm, spnn.ReLU(True)), residual blocks, the point <-> voxel helpers of
    network/utils.py, torch's own nn.Linear / BatchNorm1d / Dropout in the point branch) over this package standing in
    for torchsparse -- none of lidal_amd.network (no launch plan, no fused block, no tables built ahead) -- through
    train.py:127-140 as written there: zero_grad, forward, torch.nn.functional.cross_entropy(ignore_index=255), backward,
    Adam.  What `install_as_torchsparse()` users get; the fusions it benefits from are the ones the SURFACE carries."""
    import lidal_amd
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import surface_unet
    from lidal_amd import backend as B
    torch.manual_seed(7122)
    # the ONE line of a drop-in user.  Its default (round 6) hands torch's own Linear / BatchNorm1d / ReLU modules of the
    # model to this package at the model's first forward call; adopt=False times the opt-out
    lidal_amd.install_as_torchsparse(adopt_torch_modules=adopt)
    model = surface_unet.build(lidal_amd)[model_name](19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    coords, feats, labels = batch
    autocast = dtype == 'bf16'

    def step():
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
            logits, _ = model(lidal_amd.SparseTensor(feats, coords))
        loss = torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=255, reduction='mean')
        loss.backward()
        opt.step()
        return loss.detach()
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        B.HITS.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lidal_amd.install_as_torchsparse(adopt_torch_modules=False)      # (process-wide switch: the other legs build plain models)
    assert np.isfinite(loss.item())
    assert adopt == type(model.classifier[0]).__module__.startswith('lidal_amd')
    ms = dt / steps * 1e3
    n = int(coords.shape[0])
    return {'ms_per_step': round(ms, 3), 'voxels_per_step': n, 'voxels_per_s': round(n / ms * 1e3, 1), 'steps': steps,
            'loss': round(float(loss.item()), 4),
            'library_calls_per_step': int(sum(v for k, v in B.HITS.items() if not k.startswith('torch_fallback')) / steps),
            'what': ('surface-only %s (scripts/surface_unet.py over lidal_amd as torchsparse; the user\'s script as the reference '
                     'writes it: torch nn.Linear / BatchNorm1d / cross_entropy), per-operator path, tables built inside the '
                     'forward pass; ' % model_name)
                    + ('install_as_torchsparse() as it is by default: the point branch and the classifier are handed to this '
                       'package\'s kernels at the model\'s first forward call (same parameters, same state_dict keys; torch\'s '
                       'cross_entropy stays)' if adopt else
                       'install_as_torchsparse(adopt_torch_modules=False): torch\'s own modules stay torch\'s')}


def variant_line(res):
    ms = res['seconds'] / res['steps'] * 1e3
    return {'ms_per_step': round(ms, 3), 'voxels_per_step': int(res['voxels']),
            'voxels_per_s': round(res['voxels'] / ms * 1e3, 1), 'steps': res['steps'],
            'loss': round(res['loss'], 4)}


# ---------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# ---------------------------------------------------------------------------------------------
KERNEL_SOURCES = ('conv_img.hip', 'wgrad_dma.hip', 'bn.hip', 'voxel.hip', 'elementwise.hip', 'kmap.hip', 'sort.hip',
                  'hash.hip', 'plan.hip', 'conv.hip')


def source_digest(files):
    """sha256 over the kernel sources a counter record depends on (the record is void once they change)."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, 'lidal_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_record(suffix, sources, same_workload, key):
    """Counter traffic is measured OFFLINE (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md prescribes) and kept under profiles/.  Only THIS round's record counts, and only if it was taken
    on this exact workload AND on the kernel sources as they are now (the record stores their digest and the library
    version): a changed kernel voids it -- the line then says so instead of quoting a stale number."""
    name = 'r%02d_%s' % (ROUND, suffix)
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', name)))
    except (OSError, ValueError):
        return None, 'no counter record of this round (profiles/%s)' % name
    try:
        from lidal_amd import backend as B
        lib = rec.get('library', {})
        if not same_workload(rec['workload']):
            return None, 'profiles/%s is of another workload' % name
        if lib.get('version') != int(B.lib().lidal_version()) or lib.get('sources_sha16') != source_digest(sources):
            return None, 'profiles/%s is STALE: taken on another build of the kernels (sources changed since)' % name
        return rec[key], 'offline PMC passes: profiles/' + name
    except (KeyError, TypeError) as e:
        return None, 'profiles/%s unreadable: %r' % (name, e)


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
    from lidal_amd.nn.functional.conv import _weight_image
    x = torch.randn(n, ci, device=dev).to(dtype)
    img = _weight_image(torch.randn(27, ci, co, device=dev) * 0.02, dtype, n, 0)
    out = torch.empty((n, co), dtype=dtype, device=dev)

    def launch():
        B.check(B.lib().lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(order.table), B.ptr(order.perm),
                                               B.ptr(order.tile_masks), B.ptr(out), n, n, ci, co, 27, 0,
                                               B.dtype_code(dtype), None, None, 0, None, None, B.stream()), 'conv')
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
    # PMC-derived bytes per launch: measured OFFLINE (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on
    # `bench.py --roofline-only`, corrected as MI355X_MICROARCH.md prescribes) and kept under
    # profiles/; reported only if that record is of this exact workload
    traffic, traffic_src = pmc_record('pmc_conv_apply.json', ('conv_img.hip',),
                                      lambda wl: (wl['rows'], wl['rules'], wl['dtype']) == (n, m, args.dtype),
                                      'traffic_bytes')
    return {'bound': 'hbm', 'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
            'kernel': 'conv_lean_kernel<%s,6,192,8> through lidal_conv_apply_image (k3 s1 96->96)' % args.dtype,
            'launch_us': round(sec * 1e6, 2), 'rows': n, 'rules': m,
            'algorithmic_bytes_per_launch': int(algo_bytes),
            'mfma': {'achieved': round(flops / sec / 1e12, 3), 'peak': MFMA_PEAK_TFLOPS[args.dtype],
                     'unit': 'TFLOP/s',
                     'frac': round(flops / sec / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5)}}


# ---------------------------------------------------------------------------------------------
# per-family roofline of one whole step
# ---------------------------------------------------------------------------------------------
FAMILY_OF = {
    'lidal_conv_apply_image': 'conv_apply', 'lidal_conv_dgrad_bn_sums': 'conv_apply',
    'lidal_conv_apply_image_ws': 'conv_apply', 'lidal_conv_dgrad_bn_sums_ws': 'conv_apply',
    'lidal_conv_wgrad': 'conv_wgrad', 'lidal_conv_wgrad_streams': 'conv_wgrad',
    'lidal_conv_weight_image': 'weight_pack', 'lidal_conv_weight_image_batch': 'weight_pack',
    'lidal_conv_weight_image_pair': 'weight_pack',
    'lidal_bn_train_fwd': 'batch_norm', 'lidal_bn_train_fwd_tiles': 'batch_norm', 'lidal_bn_bwd': 'batch_norm', 'lidal_bn_bwd_tiles': 'batch_norm', 'lidal_bn_eval_fwd': 'batch_norm',
    # (the ReLU mask of a block's tail rides with the first pass of its BatchNorms' backward: both counted here)
    'lidal_add_relu_bwd_bn_sums': 'batch_norm', 'lidal_bn_bwd_from_sums': 'batch_norm',
    'lidal_add_relu_bwd_bn_tile_sums': 'batch_norm',
    'lidal_bn_fold': 'batch_norm', 'lidal_colsum': 'batch_norm',
    'lidal_wgrad_streams_build': 'kernel_maps',
    'lidal_hash': 'kernel_maps', 'lidal_kernel_hash': 'kernel_maps', 'lidal_hash_table_build': 'kernel_maps',
    'lidal_hash_table_build_coords': 'kernel_maps',
    'lidal_hash_table_query': 'kernel_maps', 'lidal_unique_sorted_i64': 'kernel_maps',
    'lidal_downsample': 'kernel_maps', 'lidal_kmap_build': 'kernel_maps', 'lidal_kmap_build_batch': 'kernel_maps', 'lidal_kmap_invert': 'kernel_maps',
    'lidal_kmap_order': 'kernel_maps', 'lidal_floor_coords': 'kernel_maps', 'lidal_revoxelize_coords': 'kernel_maps', 'lidal_kmap_order_batch': 'kernel_maps',
    'lidal_downsample_pyramid': 'kernel_maps', 'lidal_kmap_from_rules': 'kernel_maps',
    'lidal_count': 'point_voxel', 'lidal_voxelize_fwd': 'point_voxel', 'lidal_voxelize_bwd': 'point_voxel',
    'lidal_voxelize_fwd_1to1': 'point_voxel',
    'lidal_devoxelize_fwd': 'point_voxel', 'lidal_devoxelize_bwd': 'point_voxel',
    'lidal_invlist_build': 'point_voxel', 'lidal_voxelize_fwd_sorted': 'point_voxel',
    'lidal_devoxelize_bwd_sorted': 'point_voxel', 'lidal_ti_weights': 'point_voxel',
    'lidal_devoxelize_bwd_cells': 'point_voxel',
    'lidal_copy2d': 'fused_elementwise', 'lidal_add2d': 'fused_elementwise', 'lidal_transpose_f32': 'fused_elementwise',
    'lidal_cast_rows_bf16': 'fused_elementwise',
    'lidal_add_relu_fwd': 'fused_elementwise', 'lidal_add_relu_bwd': 'fused_elementwise',
    'lidal_ce_fwd': 'fused_elementwise', 'lidal_ce_bwd': 'fused_elementwise',
}


def _val(a):
    v = getattr(a, 'value', a)
    try:
        return 0 if v is None else int(v)
    except (TypeError, ValueError):
        return 0


def map_rules(coords):
    """rules per level of a coordinate set (same voxel sets whichever row order the model uses): {output rows of a
    27-offset map: its rules}, and the algorithmic bytes of building every map of the network."""
    from lidal_amd import SparseTensor
    from lidal_amd.nn.functional.conv import prefetch_kernel_maps
    from lidal_amd.network.unet import _SparseUNet
    with torch.no_grad():
        x = prefetch_kernel_maps(SparseTensor(None, coords), _SparseUNet.MAP_PLAN)
        rules = {}
        kmap_batch_bytes = 0                # every map of the network, built by ONE lidal_kmap_build_batch call
        for key, km in x.kmaps.items():
            if km.volume == 27:
                rules[km.sizes[1]] = km.total
            kmap_batch_bytes += 16 * km.sizes[0] + 16 * km.sizes[1] + 8 * km.total
    torch.cuda.synchronize()
    return rules, kmap_batch_bytes


def conv_rules(rules, k, n_in, n_out):
    """rules of a convolution call: dense layers one per row, 2x2x2 stride-2 maps one per fine row, 27-offset maps as
    measured on this coordinate set"""
    if k == 1:
        return n_out
    if k == 8:
        return max(n_in, n_out)
    return rules.get(n_out, rules.get(n_in, 6 * n_out))


def family_table(step, coords, dtype_name, step_ms):
    """Runs `step` once with every library call bracketed by events and prices each call with the
    algorithmic bytes / FLOPs of SURVEY.md 8(d):
      conv fwd / dgrad   b(N_in Ci + N_out Co) + b K Ci Co + 8 M     2 M Ci Co
      weight gradient    b(N_in Ci + N_out Co) + 4 K Ci Co + 8 M     2 M Ci Co
      BatchNorm          fwd 3 N C b, bwd 5 N C b;   relu(a+b) 3 N C b;   CE N (C b + 8)
      kernel map         16 N_in + 16 N_out + 8 M per map (hash, unique, probe, compaction, row order)
      point<->voxel      one row in + one row out per gathered row (b C each)
    M (rules) of a 27-offset map is measured on this batch's levels; 8-offset (stride 2) maps have
    M = N_fine, dense layers M = N.  Everything between library calls (Adam, cat, dropout, casts,
    host gaps) is `other` = step time - sum of the bracketed intervals."""
    from lidal_amd import SparseTensor
    from lidal_amd import backend as B
    from lidal_amd.nn.functional.conv import prefetch_kernel_maps
    from lidal_amd.network.unet import _SparseUNet
    b_el = 2 if dtype_name == 'bf16' else 4
    kmap_points = int(coords.shape[0])  # (SPVCNN's points: the voxel centres of the batch, one per voxel)
    rules, kmap_batch_bytes = map_rules(coords)
    calls = []
    # the profiled step runs operator by operator (LIDAL_PLAN=0's path): the launch plan issues the same kernels with
    # the same arguments (tests/test_plan_gpu.py: bitwise), but as words of one call that no per-call timer can bracket
    from lidal_amd.network import plan as _plan
    planned = _plan.ENABLED
    _plan.ENABLED = False
    B.set_call_timer(lambda name, a, e0, e1: calls.append((name, [_val(v) for v in a], e0, e1)))
    try:
        step()                              # (registers what the per-operator path keeps to itself)
        torch.cuda.synchronize()
        del calls[:]
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        prof_ms = (time.perf_counter() - t0) * 1e3
    finally:
        B.set_call_timer(None)
        _plan.ENABLED = planned
    fam = {}
    fwd_convs = []                      # (k, n_in, n_out, ci, co, b) of the forward convolutions, for `compulsory`
    in_backward = False

    def rules_of(k, n_in, n_out):
        return conv_rules(rules, k, n_in, n_out)

    dump = os.environ.get('BENCH_FAMILY_CALLS')          # a path: one line per library call (family, name, ms, integer arguments)
    dump_f = open(dump, 'w') if dump else None
    for name, a, e0, e1 in calls:
        f = FAMILY_OF.get(name, 'other_lib')
        ms = e0.elapsed_time(e1)
        by = fl = 0.0
        if dump_f is not None:
            dump_f.write(json.dumps({'family': f, 'name': name, 'ms': round(ms, 4),
                                     'args': [v for v in a[:16] if abs(v) < (1 << 40)]}) + '\n')
        if name in ('lidal_conv_apply_image', 'lidal_conv_dgrad_bn_sums', 'lidal_conv_apply_image_ws',
                    'lidal_conv_dgrad_bn_sums_ws'):
            n_in, n_out, ci, co, k, dt = a[6], a[7], a[8], a[9], a[10], a[12]
            b = 2 if dt == 1 else 4
            m = rules_of(k, n_in, n_out)
            by = b * (n_in * ci + n_out * co) + b * k * ci * co + 8 * m
            fl = 2.0 * m * ci * co
            if not in_backward:
                fwd_convs.append((k, n_in, n_out, ci, co, b, m))
        elif name == 'lidal_conv_wgrad':
            # (a, b, n_a, n_b, pairs, koff, a_col, gw, partial, n_slabs, k, ca, cb, dtype, stream)
            n_a, n_b, k, ca, cb, dt = a[2], a[3], a[10], a[11], a[12], a[13]
            b = 2 if dt == 1 else 4
            # a = the saved input x [n_a, ca] of the forward conv, b = grad_out [n_b, cb]
            m = rules_of(k, n_a, n_b)
            by = b * (n_a * ca + n_b * cb) + 4 * k * ca * cb + 8 * m
            fl = 2.0 * m * ca * cb
        elif name == 'lidal_conv_wgrad_streams':
            # (a, b, n_a, n_b, spairs, sdesc, n_wg, a_col, gw, partial, n_slabs, k, ca, cb, dtype, stream): the same product
            n_a, n_b, k, ca, cb, dt = a[2], a[3], a[11], a[12], a[13], a[14]
            b = 2 if dt == 1 else 4
            m = rules_of(k, n_a, n_b)
            by = b * (n_a * ca + n_b * cb) + 4 * k * ca * cb + 8 * m
            fl = 2.0 * m * ca * cb
        elif name == 'lidal_conv_weight_image':
            by = (4 + b_el) * a[5] * a[6] * a[7]
        elif name == 'lidal_conv_weight_image_pair':
            by = (4 + 2 * b_el) * a[7] * a[8] * a[9]
        elif name in ('lidal_bn_train_fwd', 'lidal_bn_eval_fwd', 'lidal_bn_train_fwd_tiles'):
            # the statistics pass that `_tiles` no longer makes stays in the algorithmic count (3 N C b)
            by = (2 if name == 'lidal_bn_eval_fwd' else 3) * a[2] * a[3] * (2 if a[1] == 1 else 4)
        elif name in ('lidal_bn_bwd', 'lidal_bn_bwd_tiles'):
            # (x, dy, dy_stride, dtype, n, c, ...): the sums pass `_tiles` no longer makes stays in the count (5 N C b)
            by = 5 * a[4] * a[5] * (2 if a[3] == 1 else 4)
        elif name == 'lidal_bn_bwd_from_sums':
            by = 5 * a[4] * a[5] * (2 if a[3] == 1 else 4)         # (as lidal_bn_bwd: the pass it no longer makes stays in the count)
        elif name in ('lidal_add_relu_bwd_bn_sums', 'lidal_add_relu_bwd_bn_tile_sums'):
            by = 3 * a[4] * a[5] * (2 if a[3] == 1 else 4)         # (as lidal_add_relu_bwd: out, g -> gm)
        elif name == 'lidal_colsum':
            by = a[2] * a[3] * (2 if a[1] == 1 else 4)
        elif name in ('lidal_add_relu_fwd', 'lidal_add_relu_bwd'):
            by = 3 * a[3] * (2 if a[4] == 1 else 4)
        elif name in ('lidal_ce_fwd', 'lidal_ce_bwd'):
            in_backward = in_backward or name == 'lidal_ce_bwd'
            by = a[3] * (a[4] * (2 if a[1] == 1 else 4) + 8) * (1 if name == 'lidal_ce_fwd' else 2)
        elif name == 'lidal_kmap_build':
            n_out, k = a[3], a[5]
            n_in = a[1] // 28 if a[1] else n_out                # table: 14 B per slot (12 + its share of the two bitmaps), 2 slots per key
            by = 16 * n_in + 16 * n_out + 8 * rules_of(k, n_in, n_out)
        elif name == 'lidal_kmap_build_batch':
            by = kmap_batch_bytes                               # (host arrays of pointers: the maps of the whole network)
        elif name == 'lidal_hash':
            by = (16 + 8) * a[1]
        elif name == 'lidal_floor_coords':
            by = (16 + 16) * a[1]
        elif name == 'lidal_revoxelize_coords':
            by = (16 + 16 + 16) * a[1]
        elif name == 'lidal_kernel_hash':
            by = 16 * a[1] + 8 * a[3] * a[1]
        elif name == 'lidal_hash_table_build':
            by = (8 + 12) * a[1]                                # key in, one slot written
        elif name == 'lidal_hash_table_build_coords':
            by = (16 + 12) * a[1]                               # coordinate row in, one slot written
        elif name == 'lidal_hash_table_query':
            by = (8 + 12 + 8) * a[3]                            # query in, one slot probed, index out
        elif name == 'lidal_unique_sorted_i64':
            by = (8 + 8) * a[1]
        elif name == 'lidal_downsample':
            by = 16 * a[1] + 16 * a[1] // 4                     # rows in, ~1/4 of them out
        elif name == 'lidal_downsample_pyramid':
            by = 16 * a[1] + 16 * a[1] // 2                     # rows in, all coarser levels out (~1/2 of them together)
        elif name == 'lidal_kmap_order_batch':
            pass                                                # (host arrays of pointers: priced with lidal_kmap_build's rules)
        elif name == 'lidal_kmap_invert':
            by = 4 * a[2] * (a[1] + a[4])                       # [k, n_out] read, [k, n_in] written
        elif name == 'lidal_kmap_order':
            by = 4 * a[2] * a[1] * 2 + 8 * a[1]                 # table read + permuted table written + perm / masks
        elif name == 'lidal_count':
            by = 4 * a[1] + 4 * a[3]
        elif name == 'lidal_ti_weights':
            by = a[3] * (16 + 64 + 32 + 32)                     # coords + idx i64 [8] in, w f32 [8] + idx i32 [8] out
        elif name == 'lidal_invlist_build':
            by = a[2] * (4 + 4 + 4) + 8 * a[3]
        elif name in ('lidal_voxelize_fwd_sorted', 'lidal_devoxelize_bwd_sorted'):
            m_rows, c, dt, n_ent = a[5], a[6], a[7], a[8]
            by = (n_ent + m_rows) * c * (2 if dt == 1 else 4)
        elif name == 'lidal_devoxelize_bwd_cells':
            # (gout, vorder, vseg, w8, corder, cseg, gin, m, c, dtype, ...): priced like the per-voxel form it replaces --
            # one row in per (point, corner), one row out per voxel -- with the points of the lists
            m_rows, c, dt = a[7], a[8], a[9]
            by = (8 * (kmap_points or m_rows) + m_rows) * c * (2 if dt == 1 else 4)
        elif name == 'lidal_voxelize_fwd_1to1':
            by = 2 * a[3] * a[4] * (2 if a[5] == 1 else 4)                          # a row in, a row out
        elif name == 'lidal_voxelize_bwd':
            by = ((2 if a[3] else 1) * a[5] + a[6]) * a[7] * (2 if a[8] == 1 else 4)     # gin (+ residual) + gout rows
        elif name == 'lidal_devoxelize_fwd':
            by = (8 * a[4] + a[4]) * a[6] * (2 if a[7] == 1 else 4)
        d = fam.setdefault(f, {'ms': 0.0, 'launches': 0, 'bytes': 0.0, 'flops': 0.0})
        d['ms'] += ms
        d['launches'] += 1
        d['bytes'] += by
        d['flops'] += fl
    if dump_f is not None:
        dump_f.close()
    out = {}
    tot_ms = tot_by = tot_fl = 0.0
    peak_tf = MFMA_PEAK_TFLOPS[dtype_name]
    for f, d in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
        tot_ms += d['ms']
        tot_by += d['bytes']
        tot_fl += d['flops']
        row = {'ms': round(d['ms'], 3), 'calls': d['launches'],
               'algorithmic_GB': round(d['bytes'] / 1e9, 3),
               'hbm_frac': round(d['bytes'] / (d['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if d['ms'] > 0 else None}
        if d['flops']:
            row['GFLOP'] = round(d['flops'] / 1e9, 1)
            row['mfma_frac'] = round(d['flops'] / (d['ms'] * 1e-3) / 1e12 / peak_tf, 4)
        out[f] = row
    # self-check of the pricing: the weight gradients do 2 M Ci Co per layer, the forward + data gradient
    # 4 M Ci Co (the stem's data gradient is not computed, the rest is) -> the ratio sits at ~0.5
    pricing_check = None
    if 'conv_apply' in fam and 'conv_wgrad' in fam and fam['conv_apply']['flops'] > 0:
        ratio = fam['conv_wgrad']['flops'] / fam['conv_apply']['flops']
        pricing_check = {'wgrad_over_fwd_plus_dgrad_flops': round(ratio, 3), 'ok': bool(0.4 <= ratio <= 0.6)}
    # COMPULSORY bytes of the step, SURVEY.md 8(d): every convolution moves each feature row once
    # forward and (2 in + 1 out) backward, its weights and rule pairs; a BatchNorm / ReLU / residual sum
    # that directly follows a convolution costs nothing extra (fusable); kernel maps, point<->voxel and
    # the loss as priced above
    comp = 0.0
    for k, n_in, n_out, ci, co, b, m in fwd_convs:
        comp += b * (n_in * ci + n_out * co) + b * k * ci * co + 8 * m                      # forward
        comp += b * (2 * n_in * ci + n_out * co + k * ci * co) + 4 * k * ci * co + 8 * m    # backward
    for f in ('kernel_maps', 'point_voxel'):
        comp += fam.get(f, {'bytes': 0.0})['bytes']
    out['other'] = {'ms': round(max(step_ms - tot_ms, 0.0), 3),
                    'what': 'torch ops between library calls (Adam, cat, dropout, casts) + gaps'}
    out['whole_step'] = {
        'ms': round(step_ms, 3), 'profiled_step_ms': round(prof_ms, 3),
        'profiled_on': 'the per-operator path (LIDAL_PLAN=0): the kernels of the planned step, one call each',
        'algorithmic_GB': round(tot_by / 1e9, 3),
        'GFLOP': round(tot_fl / 1e9, 1),
        'hbm_frac': round(tot_by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        'pricing_check': pricing_check,
        'compulsory_GB': round(comp / 1e9, 3),
        'hbm_frac_compulsory': round(comp / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        'mfma_frac': round(tot_fl / (step_ms * 1e-3) / 1e12 / peak_tf, 4),
        'ms_at_hbm_roof': round(tot_by / (HBM_PEAK_GBS * 1e9) * 1e3, 3),
        'ms_at_mfma_roof': round(tot_fl / (peak_tf * 1e12) * 1e3, 3)}
    # counter traffic of the whole step, measured OFFLINE like roofline.traffic (two --pmc passes over bench.py itself:
    # scripts/gpu/archive/r5_step_traffic.sh) and kept under profiles/; reported only for this exact workload
    tr, src = pmc_record('pmc_step_traffic.json', KERNEL_SOURCES,
                         lambda wl: (wl['rows'], wl['dtype']) == (int(coords.shape[0]), dtype_name), 'traffic_GB')
    out['whole_step']['traffic_GB'] = tr
    out['whole_step']['traffic_source'] = src
    return out


# ---------------------------------------------------------------------------------------------
# CPU baselines
# ---------------------------------------------------------------------------------------------
def cpu_baseline(args, rank_seed=7122, sample_points=None, max_threads=32, runs=5, warmups=2):
    """Oracle (CPU restatement of the torchsparse path) on a BOUNDED sample of the same workload:
    one synthetic scan of `sample_points` points (same generator and input pipeline as the bench
    batch), forward + CE + backward (upstream torchsparse has no CPU backward; autograd through the
    restatement supplies it): SURVEY.md 8d's protocol -- median of `runs` (5) after `warmups` (2), kernel
    maps rebuilt every run.  Threads = min(host cores, max_threads): the one deviation from "all cores",
    stated in `sample` (the per-offset index_select -> mm -> index_add of ~10^4 rows does not scale past
    a socket's worth of threads)."""
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
    times = []
    for _ in range(runs + warmups):
        model.zero_grad()
        t0 = time.perf_counter()
        logits, _ = model(tsref.SparseTensor(feats, coords))
        loss = torch.nn.functional.cross_entropy(logits, labels, ignore_index=255, reduction='mean')
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        times.append((t2 - t0, t1 - t0, t2 - t1))
    times = sorted(times[warmups:])
    tot, fwd, bwd = times[len(times) // 2]
    n = coords.shape[0]
    # the intra-op pool of the CPU run must not linger: its (spinning) workers compete with the one Python
    # thread that issues the GPU work -- the host-bound single-scan variant measured 2-3 ms slower after it
    torch.set_num_threads(1)
    return {'value': round(n / tot, 1), 'unit': 'voxels/s', 'cores': cores, 'kind': 'port',
            'sample': '1 synthetic scan of %d points (%d voxels), %s f32 fwd+CE+bwd on the CPU oracle: '
                      'median of %d runs after %d warm-ups (fwd %.1f s, bwd %.1f s), %d threads of %d host cores'
                      % (sample_points, n, args.model, runs, warmups, fwd, bwd, cores, os.cpu_count() or 1)}


def _score_worker(job):
    i, nei, dis = job
    from oracle import scoring_ref
    g = _score_worker.shared
    t0 = time.perf_counter()
    scoring_ref.score_frame(i, g['probs'], g['worlds'], g['sv2point'][i], nei, dis)
    return time.perf_counter() - t0


def _score_worker_init(probs, worlds, sv2point):
    _score_worker.shared = {'probs': probs, 'worlds': worlds, 'sv2point': sv2point}


def scoring_cpu_baseline(frames, nei, cores_cap=24):
    """CPU baseline of the inter-frame scoring: oracle.scoring_ref.score_frame (the restatement of
    worker_func, LiDAL.py:27-103, pinned bit for bit to the reference) under a process pool of
    min(24, cores) workers as LiDAL.py:204-206 runs it, on a BOUNDED number of frames of the same
    synthetic sequence (random but normalised probabilities: the CPU path's cost does not depend on
    the values).  The 8-view model inference that precedes it has no CPU form in the reference."""
    import multiprocessing as mp
    cores = min(cores_cap, os.cpu_count() or 1)
    rng = np.random.default_rng(3)
    probs = []
    for f in frames:
        p = rng.random((f['world'].shape[0], 19), dtype=np.float32) + 0.05
        probs.append(p / p.sum(1, keepdims=True))
    worlds = [f['world'] for f in frames]
    sv = [f['sv2point'] for f in frames]
    n_jobs = min(len(frames), cores)
    ctx = mp.get_context('spawn')                       # never fork a process that has initialised the GPU
    with ctx.Pool(cores, initializer=_score_worker_init, initargs=(probs, worlds, sv)) as pool:
        pool.map(_score_worker, [(0, nei, 0.1)])        # warm-up: imports, page-in
        t0 = time.perf_counter()
        per = pool.map(_score_worker, [(i, nei, 0.1) for i in range(n_jobs)], chunksize=1)
        dt = time.perf_counter() - t0
    return {'value': round(n_jobs / dt, 3), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d frames of %d points, nei_num %d, oracle.scoring_ref.score_frame (worker_func '
                      'restatement, KD-tree queries) under Pool(%d): %.1f s wall, %.1f s per frame per worker; '
                      'scoring only (the reference has no CPU inference path)'
                      % (n_jobs, frames[0]['world'].shape[0], nei, cores, dt, float(np.mean(per)))}


# ---------------------------------------------------------------------------------------------
# secondary metric: prob_inference + LiDAL scoring
# ---------------------------------------------------------------------------------------------
def _gen_frame_block(job):
    n, points, seed, start, total = job
    from lidal_amd import synth
    return synth.make_sequence(n, n_points=points, seed=seed, start=start, total=total)


def _gen_score_batch(job):
    pts, inten, seed = job
    from lidal_amd import synth
    return synth.make_score_batch(pts, inten, np.random.default_rng(seed), inf_reps=8)


def make_scoring_inputs(args, world, rank):
    """Synthetic frames of this rank's block + their 8 augmented views, generated by a small spawn
    pool (32 frames x 8 views of 120 k points is ~50 s of single-core numpy otherwise)."""
    import multiprocessing as mp
    per = args.score_frames
    total = per * world
    workers = max(1, min(8, (os.cpu_count() or 1) // max(world, 1), per))
    if workers == 1 or per < 4:
        frames = _gen_frame_block((per, args.points, 7122, rank * per, total))
        batches = [_gen_score_batch((f['points'], f['intensity'], [7122, 99, rank, i]))
                   for i, f in enumerate(frames)]
        return frames, batches
    ctx = mp.get_context('spawn')
    with ctx.Pool(workers) as pool:
        bounds = np.linspace(0, per, workers + 1).astype(int)
        jobs = [(int(bounds[i + 1] - bounds[i]), args.points, 7122, rank * per + int(bounds[i]), total)
                for i in range(workers) if bounds[i + 1] > bounds[i]]
        frames = [f for blk in pool.map(_gen_frame_block, jobs) for f in blk]
        batches = pool.map(_gen_score_batch, [(f['points'], f['intensity'], [7122, 99, rank, i])
                                              for i, f in enumerate(frames)], chunksize=1)
    return frames, batches


def bench_scoring(args, model, world, rank, dev, frames, batches):
    """prob_inference (8 augmented views per frame) + inter-frame scoring + the return leg to rank
    0, frames sharded over ranks in the reference's contiguous blocks; probabilities / world
    coordinates exchanged by one all_gather_into_tensor each."""
    from lidal_amd.score import collect_sequence, interframe, score_sequence
    per = args.score_frames
    total = per * world
    dev_frames = []
    for f, sb in zip(frames, batches):
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], dev)
        dev_frames.append({'coords': torch.from_numpy(sb['coords_v_b']).to(dev),
                           'feats': torch.from_numpy(sb['feats_v_b']).to(dev),
                           'inverse': torch.from_numpy(sb['inverse_indices_b']).to(dev),
                           'world': torch.from_numpy(f['world']).to(dev), 'sv_ptr': ptr, 'sv_idx': idx})
    sv_ids = [f['sv_id'] for f in frames]
    if os.environ.get('BENCH_EMPTY_CACHE'):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    model.eval()
    log('scoring inputs resident')
    # The reference infers in fp32 (score/prob_inference.py:91-113: no autocast anywhere in the repository) and
    # BASELINE.json grants bf16 to configs[1] (the train step) only: the HEADLINE frames/s is the f32 one.  The bf16
    # figure (bf16 conv operands at inference; scoring arithmetic unchanged) is reported beside it under `by_dtype`;
    # tests/test_benchsize_gpu.py bounds what it does to the scores and the selected flags.
    out = {'metric': 'frames/sec prob_inference(8 views)+LiDAL scoring', 'unit': 'frames/s', 'dtype': 'f32',
           'frames': total, 'frames_per_rank': per, 'points_per_frame': args.points,
           'voxels_per_frame': int(np.mean([d['coords'].shape[0] for d in dev_frames])),
           'exchange': ('halo exchange (batch_isend_irecv of the prob f32 [P,19] / world f64 [P,3] frames other ranks read) '
                        '+ sv results to rank 0' if world > 1 else 'none (1 rank)'), 'by_nei': {}, 'by_dtype': {}}

    def run(nei, autocast):
        scores = score_sequence(model, dev_frames, rank * per, total, nei_num=nei, dis_thresh=0.1,
                                inf_reps=8, autocast=autocast)
        return scores, collect_sequence(scores, sv_ids, [d['sv_ptr'] for d in dev_frames], rank * per, total)
    for dtype_name in (['f32'] if os.environ.get('BENCH_SECONDARY_F32_ONLY') else ['f32', 'bf16']):
        autocast = dtype_name == 'bf16'
        by_nei = {}
        for nei in args.nei:
            if total < nei + 3:
                continue
            run(nei, autocast)                      # warm-up
            barrier_sync(world)
            t0 = time.perf_counter()
            scores, got = run(nei, autocast)
            barrier_sync(world)
            dt = max_over_ranks(time.perf_counter() - t0, world, dev)
            assert all(torch.isfinite(o[0]).all() for o in scores)
            assert (got is not None) == (rank == 0)
            by_nei[str(nei)] = {'value': round(total / dt, 3), 'ms_per_frame_per_gpu': round(dt / per * 1e3, 3)}
        out['by_dtype'][dtype_name] = {'by_nei': by_nei,
                                       'what': ('f32 features, weights and accumulation (the reference\'s precision, prob_inference.py:91-113); '
                                                'products in the split form, LIDAL_F32_SPLIT=1: each operand cut exactly into 3 bf16 '
                                                'pieces, 6 partial products on the bf16 MFMA (1.3e-6 of f64 on the 96->96 layer against '
                                                '1.5e-6 for the exact f32 MFMA); the exact-f32 figure: by_dtype.f32_exact'
                                                if dtype_name == 'f32' else
                                                'bf16 conv operands / f32 accumulation at inference; scoring arithmetic as f32')}
    out['by_nei'] = out['by_dtype']['f32']['by_nei']
    first = str(args.nei[0]) if str(args.nei[0]) in out['by_nei'] else next(iter(out['by_nei']), None)
    if first is not None:
        out['value'] = out['by_nei'][first]['value']
        out['ms_per_frame_per_gpu'] = out['by_nei'][first]['ms_per_frame_per_gpu']
        out['nei_num'] = int(first)
        if world == 1:
            # the same frames on the EXACT f32 MFMA (LIDAL_F32_SPLIT=0), so that the headline's precision label is auditable
            from lidal_amd import backend as B
            split = B.SPLIT_F32
            B.SPLIT_F32 = False
            try:
                run(int(first), False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(int(first), False)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                out['by_dtype']['f32_exact'] = {
                    'by_nei': {first: {'value': round(total / dt, 3), 'ms_per_frame_per_gpu': round(dt / per * 1e3, 3)}},
                    'what': 'LIDAL_F32_SPLIT=0: every product on v_mfma_f32_16x16x4_f32 (157 TFLOP/s dense peak)'}
            finally:
                B.SPLIT_F32 = split
            # frames/s as a fraction of the roofline (north_star): the frame's convolution FLOPs over the whole-job time per
            # frame against the MFMA bound of the arithmetic actually used, the scorer's bytes over its kernel time against HBM
            out['roofline'] = guarded(scoring_roofline, model, dev_frames, int(first), 'f32',
                                      out['ms_per_frame_per_gpu'])
            if 'bf16' in out['by_dtype'] and first in out['by_dtype']['bf16']['by_nei']:
                out['by_dtype']['bf16']['roofline'] = guarded(
                    scoring_roofline, model, dev_frames, int(first), 'bf16',
                    out['by_dtype']['bf16']['by_nei'][first]['ms_per_frame_per_gpu'])
    model.train()
    return out


def scoring_roofline(model, dev_frames, nei, dtype_name, ms_per_frame):
    """One frame of the secondary metric call by call (the per-operator path: a launch plan is one call no timer can
    bracket -- same kernels, tests/test_plan_gpu.py): the convolution / dense-layer FLOPs of its 8-view inference
    (2 M Ci Co per call, M = the rules of the call's kernel map as measured on this frame) and the algorithmic bytes of
    scoring it ((1 + nei) P (C 4 + 24) for the inter-frame kernels, the NN grid of ONE frame -- every
    frame's grid is built once and read by its nei neighbours --, the supervoxel reduction).
      inference   FLOPs per frame / the WHOLE-JOB time per frame (1 / frames-per-second) against the MFMA bound of the
                  arithmetic actually used: 2.5 PFLOP/s bf16; a sixth of it for f32 in the split form (six bf16 MFMAs
                  per f32 product block; the 4-channel stem runs on the exact f32 MFMA and is priced with the rest);
                  `conv_kernels` = the same FLOPs over the bracketed time of those calls alone
      scorer      algorithmic bytes per scored frame / the bracketed time of its calls against the HBM peak"""
    from lidal_amd import backend as B
    from lidal_amd.network import plan as _plan
    from lidal_amd.score.interframe import FrameBank, score_frame
    from lidal_amd.score.prob_inference import infer_frame
    autocast = dtype_name == 'bf16'
    d = dev_frames[0]
    rules, _ = map_rules(d['coords'])
    calls = []
    planned = _plan.ENABLED
    _plan.ENABLED = False
    probs = []
    try:
        for j in range(nei + 1):          # the probabilities of one window (untimed), then frame 0 call by call
            f = dev_frames[j]
            probs.append(infer_frame(model, f['coords'], f['feats'], f['inverse'], 8, autocast=autocast)[0])
        torch.cuda.synchronize()
        B.set_call_timer(lambda name, a, e0, e1: calls.append((name, [_val(v) for v in a], e0, e1)))
        infer_frame(model, d['coords'], d['feats'], d['inverse'], 8, autocast=autocast)
        torch.cuda.synchronize()
        inf_calls = list(calls)
        del calls[:]
        bank = FrameBank(0.1, n_frames=nei + 1)
        for j in range(nei + 1):
            bank.add(dev_frames[j]['world'], probs[j], frame_id=j)
        mid = nei // 2
        B.set_call_timer(None)
        score_frame(bank, mid, dev_frames[mid]['sv_ptr'], dev_frames[mid]['sv_idx'], nei)       # (builds the grids)
        bank._grid[0] = None                                                                   # one grid build, timed
        torch.cuda.synchronize()
        B.set_call_timer(lambda name, a, e0, e1: calls.append((name, [_val(v) for v in a], e0, e1)))
        score_frame(bank, mid, dev_frames[mid]['sv_ptr'], dev_frames[mid]['sv_idx'], nei)
        torch.cuda.synchronize()
        score_calls = list(calls)
    finally:
        B.set_call_timer(None)
        _plan.ENABLED = planned
    flops = conv_ms = 0.0
    codes = {}
    for name, a, e0, e1 in inf_calls:
        if name in ('lidal_conv_apply_image', 'lidal_conv_apply_image_ws'):
            n_in, n_out, ci, co, k, dt = a[6], a[7], a[8], a[9], a[10], a[12]
            flops += 2.0 * conv_rules(rules, k, n_in, n_out) * ci * co
            conv_ms += e0.elapsed_time(e1)
            codes[dt] = codes.get(dt, 0) + 1
    form = ('bf16' if autocast else ('f32 in the split form (3 bf16 pieces per operand, 6 bf16 MFMAs per product block)'
                                     if codes.get(B.F32_SPLIT, 0) else 'exact f32 MFMA'))
    peak = MFMA_PEAK_TFLOPS['bf16'] if autocast else (SPLIT_PEAK_TFLOPS if codes.get(B.F32_SPLIT, 0) else MFMA_PEAK_TFLOPS['f32'])
    sc_bytes = sc_ms = 0.0
    for name, a, e0, e1 in score_calls:
        ms = e0.elapsed_time(e1)
        if name in ('lidal_interframe_score', 'lidal_interframe_score_ordered'):
            p, c, n_nei = a[2], a[3], a[8]
            sc_bytes += (1 + n_nei) * p * (c * 4 + 24) + p * 16
            sc_ms += ms
        elif name == 'lidal_nn_grid_build':
            p = a[1]
            sc_bytes += 24 * p + int(B.lib().lidal_nn_grid_bytes(p))
            sc_ms += ms
        elif name == 'lidal_supervoxel_reduce':
            p = int(d['world'].shape[0])
            sc_bytes += p * (8 + 4 + 24 + 8)
            sc_ms += ms
    tf = flops / (ms_per_frame * 1e-3) / 1e12
    out = {'inference': {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                         'frac': round(tf / peak, 4), 'GFLOP_per_frame': round(flops / 1e9, 1), 'arithmetic': form,
                         'time': 'whole job: %.3f ms per frame (1 / frames per second)' % ms_per_frame,
                         'conv_kernels': {'ms_per_frame': round(conv_ms, 3), 'calls': sum(codes.values()),
                                          'achieved': round(flops / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
                                          'frac': round(flops / (conv_ms * 1e-3) / 1e12 / peak, 4) if conv_ms else None}}}
    if sc_ms:
        gbs = sc_bytes / (sc_ms * 1e-3) / 1e9
        out['scorer'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(gbs / HBM_PEAK_GBS, 4), 'algorithmic_MB_per_frame': round(sc_bytes / 1e6, 2),
                         'kernel_us_per_frame': round(sc_ms * 1e3, 1), 'nei_num': nei,
                         'what': 'lidal_interframe_score + one lidal_nn_grid_build + lidal_supervoxel_reduce, HIP events around each call'}
    return out


def make_sequence_frames(n_frames, points, seed=7122):
    """The raw frames of ONE sequence of n_frames (BASELINE.json configs[3]: 256), generated by a spawn pool BEFORE this
    process touches the GPU."""
    import multiprocessing as mp
    workers = max(1, min(16, os.cpu_count() or 1, n_frames // 8))
    bounds = np.linspace(0, n_frames, workers + 1).astype(int)
    jobs = [(int(bounds[i + 1] - bounds[i]), points, seed, int(bounds[i]), n_frames) for i in range(workers)
            if bounds[i + 1] > bounds[i]]
    if workers == 1:
        return _gen_frame_block(jobs[0])
    with mp.get_context('spawn').Pool(workers) as pool:
        return [f for blk in pool.map(_gen_frame_block, jobs) for f in blk]


def bench_score_sequence(args, model, dev, frames, nei=10):
    """BASELINE.json configs[3] at its stated length on ONE GPU: a whole 256-frame sequence -- 8 augmented views per frame
    voxelised and collated on the GPU (lidal_voxelize_points, outside the timed region), f32 inference, every frame scored
    against its +-nei/2 window with the wrap rules at both ends (LiDAL.py:41-42), the FrameBank holding all 256 frames."""
    from lidal_amd import data
    from lidal_amd.score import interframe, score_sequence
    aug = np.random.RandomState(7122)
    dev_frames = []
    for f in frames:
        pts, inten = torch.from_numpy(f['points']).to(dev), torch.from_numpy(f['intensity']).to(dev)
        samples = []
        for _ in range(8):
            trans_m, rnd = data.draw_augmentation(aug)
            coords_v, feats_v, _, inverse = data.voxelize_scan(pts, inten, trans_m, rnd)
            samples.append({'coords_v': coords_v, 'feats_v': feats_v, 'inverse_idxs': inverse})
        b = data.collate(samples)
        ptr, idx, _ = interframe.sv_csr(f['sv2point'], dev)
        dev_frames.append({'coords': b['coords_v_b'].contiguous(), 'feats': b['feats_v_b'].contiguous(),
                           'inverse': b['inverse_indices_b'].contiguous(),
                           'world': torch.from_numpy(f['world']).to(dev), 'sv_ptr': ptr, 'sv_idx': idx})
    n = len(dev_frames)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    resident = int(torch.cuda.memory_allocated(dev))
    model.eval()
    try:
        score_sequence(model, dev_frames[:nei + 6], 0, nei + 6, nei_num=nei, dis_thresh=0.1, inf_reps=8)      # warm-up
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        t0 = time.perf_counter()
        scores = score_sequence(model, dev_frames, 0, n, nei_num=nei, dis_thresh=0.1, inf_reps=8)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        model.train()
    assert len(scores) == n and all(torch.isfinite(o[0]).all() and torch.isfinite(o[1]).all() for o in scores)
    return {'frames': n, 'frames_per_s': round(n / dt, 3), 'ms_per_frame': round(dt / n * 1e3, 3), 'seconds': round(dt, 3),
            'dtype': 'f32', 'nei_num': nei, 'points_per_frame': args.points,
            'voxels_per_frame': int(np.mean([d['coords'].shape[0] for d in dev_frames])),
            'inputs_resident_MB': resident >> 20,
            'peak_allocated_MB': int(torch.cuda.max_memory_allocated(dev)) >> 20,
            'peak_reserved_MB': int(torch.cuda.max_memory_reserved(dev)) >> 20,
            'what': 'one whole sequence on one GPU (config 4 at its stated length): inputs resident, 8-view f32 inference '
                    '+ inter-frame scoring of every frame, windows wrapping at both ends (LiDAL.py:41-42); the first '
                    'pass over the full length is the timed one (warm-up: %d frames)' % (nei + 6)}


def guarded(fn, *a):
    """The extras of the line (roofline, families, variants, secondary) never cost the contract fields."""
    try:
        return fn(*a)
    except Exception as e:              # noqa: BLE001
        return {'error': repr(e)}


def run_variants(args, batch, dev, inline=None):
    var = {}
    if inline is not None:
        var['inline_geometry'] = variant_line(inline)
        var['inline_geometry']['what'] = ('the same step with the coordinate tables built inside the forward pass '
                                          '(no second stream): the order of work of the reference and of rounds 1-3a')
    one = make_batch(1, args.points, 7122, dev)
    res1 = bench_train(1, 0, dev, args.model, args.dtype, one, max(args.steps, 10), 3, ddp=False)
    var['single_scan'] = variant_line(res1)
    # The single scan is host-bound (~6 ms of Python and ~800 launches per step, profiles/README.md round 6): one window of 20
    # steps moved between 5.9 and 9.0 ms from run to run on the same box.  Four more windows of the same length follow the
    # first; the figure of the leg is the MEDIAN window, every window is listed.
    wins = [var['single_scan']['ms_per_step']]
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(res1['steps']):
            res1['step']()
        torch.cuda.synchronize()
        wins.append(round((time.perf_counter() - t0) / res1['steps'] * 1e3, 3))
    med = sorted(wins)[len(wins) // 2]
    var['single_scan'].update({'ms_per_step': med, 'voxels_per_s': round(res1['voxels'] / med * 1e3, 1), 'windows_ms': wins,
                               'what': 'BASELINE.json\'s literal "@120k pts": one scan per step; median of five windows of %d steps'
                                       % res1['steps']})
    del res1
    var['single_scan']['host'] = guarded(host_calls, args.model, args.dtype, one, dev)
    # the reference draws a new augmentation per iteration (sk_dataset.py:143-171): 8 differently
    # augmented batches of the same scans, voxelised on the GPU outside the timed region, one per step
    fresh = make_fresh_batches(args.frames, args.points, 7122, dev, 8)
    # (two untimed cycles through the 8 batches: the allocator's pools then hold blocks of every size the cycle asks for)
    var['fresh_coords'] = variant_line(bench_train(1, 0, dev, args.model, args.dtype, fresh,
                                                   max(args.steps, 16), 16, ddp=False))
    var['fresh_coords']['voxels_per_batch'] = [int(b[0].shape[0]) for b in fresh]
    del fresh
    var['fresh_stream'] = guarded(bench_fresh_stream, dev, args.model, args.dtype, args.frames, args.points,
                                  max(2 * args.steps, 40))
    # (8 warm-up steps each: the first per-operator model of a process also pays the allocator's growth and the look-back's
    #  lazy first pass -- with 3, whichever of the two legs ran first measured 3-4 ms slower)
    var['dropin_surface_no_adopt'] = guarded(bench_dropin_surface, dev, args.model, args.dtype, batch, args.steps, 8, False)
    var['dropin_surface'] = guarded(bench_dropin_surface, dev, args.model, args.dtype, batch, args.steps, 8, True)
    other_dtype = 'f32' if args.dtype == 'bf16' else 'bf16'
    var[other_dtype] = variant_line(bench_train(1, 0, dev, args.model, other_dtype, batch,
                                                max(3, args.steps // 2), 2, ddp=False))
    if other_dtype == 'f32':
        from lidal_amd import backend as B
        var['f32']['what'] = ('the reference\'s training precision (train.py:127-140, no autocast): f32 features, weights, '
                              'accumulation, BatchNorm in f64 sums; forward products and data gradients of the sparse '
                              'convolutions and every weight gradient in the split form (3 exact bf16 pieces per operand, 6 partial '
                              'products on the bf16 MFMA: as close to f64 as the exact kernels); the dense layers\' products and the '
                              '4-channel stem on the exact f32 MFMA; one stream (network/plan.py SIDE_F32)'
                              if B.SPLIT_F32_TRAIN and B.SPLIT_F32 else 'every product on the exact f32 MFMA')
        if B.SPLIT_F32_TRAIN and B.SPLIT_F32:
            B.SPLIT_F32_TRAIN = False
            try:
                var['f32_exact'] = variant_line(bench_train(1, 0, dev, args.model, 'f32', batch, max(3, args.steps // 2), 2, ddp=False))
                var['f32_exact']['what'] = 'LIDAL_F32_SPLIT_TRAIN=0: every product of the f32 step on v_mfma_f32_16x16x4_f32'
            finally:
                B.SPLIT_F32_TRAIN = True
    other_model = 'minkunet' if args.model == 'spvcnn' else 'spvcnn'
    var[other_model] = variant_line(bench_train(1, 0, dev, other_model, args.dtype, batch,
                                                args.steps, 3, ddp=False))
    return var


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as the reference's entry points do
    (train.py:163-203, score/prob_inference.py:219-223: mp.spawn of one process per GPU from a plain `python train.py`).
    The parent never touches the GPU: it starts N fresh interpreters of this file (subprocess.Popen -- no fork of, and no
    exec from, a GPU-initialised process) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank
    0's ONE line and exits non-zero if any rank fails or the job times out (no retry in place)."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()                          # rank 0 writes only its ONE line there (all else goes to stderr)
    deadline = time.monotonic() + float(os.environ.get('BENCH_LAUNCH_TIMEOUT', '3000'))
    failed = None
    try:
        while failed is None and any(p.poll() is None for p in procs):
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if bad:
                failed = 'rank %d exited with code %d' % bad[0]
            elif time.monotonic() > deadline:
                failed = 'timed out'
            else:
                time.sleep(0.2)
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if failed is None and bad:
            failed = 'rank %d exited with code %d' % bad[0]
    finally:
        for p in procs:
            if p.poll() is None:                 # exactly the process groups started above
                try:
                    os.killpg(p.pid, 15)
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
    reader.join(timeout=10)
    line = b''.join(c for c in chunks if c)
    text = line.decode(errors='replace').strip()
    if failed is not None or not text:
        print('[bench] %d-rank launch failed: %s' % (n, failed or 'rank 0 printed nothing'), file=sys.stderr, flush=True)
        sys.exit(1)
    print(text, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)           # (before anything touches the GPU)
    # the contract is ONE JSON line on stdout: libraries that print there on their own (RCCL's version banner at the
    # first collective) are sent to stderr -- file descriptor 1 becomes stderr, the line goes to the original stdout
    out = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    solo = rank == 0 and world == 1         # the extras describe one GPU; N>1 runs report the contract line
    # ---- host-only work first: process pools (input generation, the CPU baselines) start their
    #      workers BEFORE this process touches the GPU (a GPU-initialised process must not exec)
    frames = batches = None
    cpu_lines = {}
    if not args.no_secondary and not args.roofline_only:
        frames, batches = make_scoring_inputs(args, world, rank)
        log('scoring inputs generated')
        if solo and not args.no_cpu_baseline:
            cpu_lines['secondary'] = guarded(scoring_cpu_baseline, frames, args.nei[0])
            log('scoring cpu baseline', cpu_lines['secondary'])
    if solo and not args.no_cpu_baseline and not args.roofline_only:
        cpu_lines['train'] = guarded(cpu_baseline, args)
        log('train cpu baseline', cpu_lines['train'])
    seq_frames = None
    if (solo and not args.no_variants and not args.no_secondary and not args.roofline_only and args.sequence_frames > 0):
        seq_frames = guarded(make_sequence_frames, args.sequence_frames, args.points)
        log('sequence of %d frames generated' % args.sequence_frames if isinstance(seq_frames, list) else seq_frames)
    # ---- GPU
    world, rank, dev = dist_setup(args)
    from lidal_amd import backend
    backend.lib()                           # fail loudly if the HIP library is missing
    if not os.environ.get('BENCH_NO_AFFINITY'):
        near = backend.bind_cpus_near(dev.index)        # (a launcher's `numactl --cpunodebind`: one process per GPU)
        log('host threads bound to the GPU\'s NUMA node:', 'no (topology not readable)' if near is None else '%d cpus' % len(near))
    batch = make_batch(args.frames, args.points, 7122 + rank, dev)
    log('batch built', tuple(batch[0].shape))
    if args.roofline_only:
        print(json.dumps({'roofline': roofline_conv(args, batch[0], dev)}), file=out, flush=True)
        return
    res = bench_train(world, rank, dev, args.model, args.dtype, batch, args.steps, args.warmup)
    log('train timed: %.3f s for %d steps' % (res['seconds'], args.steps))
    ms = res['seconds'] / args.steps * 1e3
    voxels = sum_over_ranks(float(res['voxels']), world, dev)
    line = {
        'metric': 'voxels/sec SPVCNN fwd+bwd @120k pts' if args.model == 'spvcnn'
                  else 'voxels/sec MinkUNet fwd+bwd @120k pts',
        'value': round(voxels * args.steps / res['seconds'], 1),
        'unit': 'voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s train step (train.py:127-140: fwd + CE + bwd + Adam), %d scans x ~%dk pts '
                               'per GPU, 0.05 m voxels, kernel maps rebuilt every step (each step builds the '
                               'next step\'s coordinate tables on a second stream; variants.inline_geometry: '
                               'built inside the forward pass)'
                               % (args.model, args.frames, args.points // 1000),
                   'voxels_per_step_per_gpu': int(voxels / world),
                   'parallelism': 'dp%d' % world, 'loss': round(res['loss'], 4)},
    }
    if solo:
        line['host'] = guarded(host_calls, args.model, args.dtype, batch, dev)
        if isinstance(line['host'], dict):
            # wall time the host needed to QUEUE the timed steps (no synchronisation inside).  Well under ms_per_step: the
            # host runs ahead and the GPU is the bound.  Close to it: EITHER the step is host-bound OR the host ran into the
            # depth of the hardware queues (~1.5 steps of dispatches) and waited there -- compare with the single-scan
            # variant, whose host work is the same (6.4 ms per step covers it)
            line['host']['queueing_ms_per_step'] = round(res['host_seconds'] / args.steps * 1e3, 3)
    if rank == 0 and not args.no_roofline:
        line['roofline'] = guarded(roofline_conv, args, batch[0], dev)
        log('roofline', line['roofline'])
    inline = None
    if solo and not (args.no_families and args.no_variants):
        # the same step with its tables built in line: the serial cost of every family (on two streams the
        # bracketed intervals overlap and would not add up), and the A/B of the second stream
        inline = guarded(bench_train, 1, 0, dev, args.model, args.dtype, batch, args.steps, args.warmup, False, False)
        if 'error' in inline:
            log('inline step failed', inline)
            inline = None
    if solo and not args.no_families and inline is not None:
        inline_ms = inline['seconds'] / inline['steps'] * 1e3
        line['families'] = guarded(family_table, inline['step'], batch[0], args.dtype, inline_ms)
        if isinstance(line['families'], dict) and 'whole_step' in line['families']:
            line['families']['whole_step']['what'] = ('the step with its coordinate tables built in line (%.3f ms); '
                                                      'the headline step builds them on a second stream' % inline_ms)
        log('families', line['families'])
    if solo and not args.no_variants:
        line['variants'] = guarded(run_variants, args, batch, dev, inline)
        log('variants', line['variants'])
    if frames is not None:
        sec = (guarded(bench_scoring, args, res['model'], world, rank, dev, frames, batches) if world == 1
               else bench_scoring(args, res['model'], world, rank, dev, frames, batches))
        log('secondary', sec)
        if rank == 0:
            line['secondary'] = sec
            if 'secondary' in cpu_lines:
                line['secondary']['cpu_baseline'] = cpu_lines['secondary']
    if isinstance(seq_frames, list):
        if os.environ.get('BENCH_EMPTY_CACHE', '1') != '0':
            frames = batches = None
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        leg = guarded(bench_score_sequence, args, res['model'], dev, seq_frames, args.nei[0])
        log('score_%d' % args.sequence_frames, leg)
        if isinstance(line.get('variants'), dict):
            line['variants']['score_%d' % args.sequence_frames] = leg
    if 'train' in cpu_lines:
        line['cpu_baseline'] = cpu_lines['train']
    if rank == 0:
        print(json.dumps(line), file=out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
