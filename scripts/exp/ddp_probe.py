"""Where a DDP step spends its time (two ranks on ONE device over gloo, or N ranks over RCCL): phases of train.py:127-140
timed separately (with a device synchronisation after each, so the sum is longer than a real step).
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      scripts/exp/ddp_probe.py          (BENCH_BACKEND=gloo|nccl, LIDAL_PLAN=0|1, FRAMES)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from lidal_amd import SparseTensor, synth  # noqa: E402
from lidal_amd.network import SPVCNN, GeometryPrefetcher  # noqa: E402
from lidal_amd.nn.functional.fused import cross_entropy  # noqa: E402

rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
backend = os.environ.get('BENCH_BACKEND', 'gloo')
dev = torch.device('cuda', 0 if backend == 'gloo' else int(os.environ.get('LOCAL_RANK', 0)))
torch.cuda.set_device(dev)
if world > 1 or os.environ.get('DP'):
    dist.init_process_group(backend, rank=rank, world_size=world)
b = synth.make_train_batch(n_frames=int(os.environ.get('FRAMES', '5')), n_points=120000, seed=7122 + rank)
coords, feats, labels = (torch.from_numpy(b[k]).to(dev) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
torch.manual_seed(7122)
model = SPVCNN(19).to(dev).train()
if os.environ.get('DP'):
    from lidal_amd.data_parallel import DataParallel
    net = DataParallel(model)
elif world > 1 and not os.environ.get('NO_DDP'):
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index])
else:
    net = model
opt = torch.optim.Adam(net.parameters(), fused=True)
pf = GeometryPrefetcher(model, device=dev)
g = pf.submit(coords)


def tick(names, t=[0.0]):
    torch.cuda.synchronize()
    now = time.perf_counter()
    if names is not None:
        names.append(now - t[0])
    t[0] = now


for it in range(6):
    ph = []
    tick(None)
    opt.zero_grad(); tick(ph)
    x = SparseTensor(feats, coords); x.geometry = g
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits, _ = net(x)
    tick(ph)
    loss = cross_entropy(logits, labels, ignore_index=255); tick(ph)
    if it == 3 and os.environ.get('PROFILE') and rank == 0:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            loss.backward()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=25, max_name_column_width=60), flush=True)
    else:
        loss.backward()
    tick(ph)
    opt.step(); tick(ph)
    g = pf.submit(coords); tick(ph)
    if rank == 0:
        print('step %d  zero %.1f  forward %.1f  loss %.1f  backward %.1f  adam %.1f  submit %.1f  ms   (loss %.4f)'
              % ((it,) + tuple(1e3 * v for v in ph) + (loss.item(),)), flush=True)
pf.drain()
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
