"""The level-0 96 -> 96 layer (forward convolution + weight gradient) on SPVCNN's OWN level-0 voxel order --
sorted coordinate hashes (LIDAL_L0_ORDER=hash: the reference's torch.unique order, network/utils.py:18) or a
Z-order curve (LIDAL_L0_ORDER=morton, an experiment of network/glue.py) -- for rocprofv3 (--kernel-trace --stats and
--pmc FETCH_SIZE / WRITE_SIZE passes: scripts/exp/l0_order.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lidal_amd import backend as B, synth  # noqa: E402
if os.environ.get('LIDAL_L0_ORDER') == 'morton':
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import l0_morton  # noqa: E402,F401
from lidal_amd.network import SPVCNN, Geometry  # noqa: E402
from lidal_amd.nn.functional.conv import _weight_image, wgrad_scratch  # noqa: E402

dev = torch.device('cuda', 0)
b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).train()
g = Geometry.build(model, coords, True)
km = g.x0.kmaps[((1, 1, 1), (3, 3, 3), (1, 1, 1), (1, 1, 1))]
n = km.sizes[0]
order = km.order_out
x = torch.randn(n, 96, device=dev).bfloat16()
gy = torch.randn(n, 96, device=dev).bfloat16()
img = _weight_image(torch.randn(27, 96, 96, device=dev) * 0.02, torch.bfloat16, n, 0)
out = torch.empty((n, 96), dtype=torch.bfloat16, device=dev)
gw = torch.empty((27, 96, 96), dtype=torch.float32, device=dev)
partial = wgrad_scratch(n, n, 27, 96, 96, torch.bfloat16, dev)
L = B.lib()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
reps = int(os.environ.get('REPS', '10'))
for it in range(2):
    ev[0].record()
    for _ in range(reps):
        B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(order.table), B.ptr(order.perm), B.ptr(order.tile_masks),
                                         B.ptr(out), n, n, 96, 96, 27, 0, B.BF16, None, None, 0, None, None, B.stream()), 'conv')
    ev[1].record()
    for _ in range(reps):
        B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(km._nbmaps_cap), B.ptr(km.koff), 0, B.ptr(gw), B.ptr(partial),
                                   partial.shape[0], 27, 96, 96, B.BF16, B.stream()), 'wgrad')
    ev[2].record()
    torch.cuda.synchronize()
print(os.environ.get('LIDAL_L0_ORDER', 'hash'), 'rows', n, 'rules', km.total,
      'conv_apply %.1f us  wgrad (+reduce) %.1f us' % (ev[0].elapsed_time(ev[1]) * 1e3 / reps, ev[1].elapsed_time(ev[2]) * 1e3 / reps))
