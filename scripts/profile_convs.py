"""GPU: every convolution / weight-gradient launch of one bench train step with its shape and duration
(events on the launch stream around each library call), sorted by time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from lidal_amd import backend as B
    from lidal_amd.network import SPVCNN
    from lidal_amd.train_step import train_step
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device('cuda')
    coords, feats, labels = bench.make_batch(frames, 120000, 7122, dev)
    torch.manual_seed(7122)
    model = SPVCNN(19).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), fused=True)
    for _ in range(3):
        train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    calls = []
    B.set_call_timer(lambda name, a, e0, e1: calls.append((name, [bench._val(v) for v in a], e0, e1)))
    train_step(model, opt, feats, coords, labels, autocast=True)
    torch.cuda.synchronize()
    B.set_call_timer(None)
    rows = {}
    for name, a, e0, e1 in calls:
        ms = e0.elapsed_time(e1)
        if name in ('lidal_conv_apply_image', 'lidal_conv_dgrad_bn_sums'):
            key = ('apply', a[6], a[7], a[8], a[9], a[10], a[11])          # n_in n_out ci co k kflip
        elif name == 'lidal_conv_wgrad':
            key = ('wgrad', a[2], a[3], a[11], a[12], a[10], a[6])
        else:
            continue
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += ms
    tot = sum(r[1] for r in rows.values())
    print('conv launches of one step: %.3f ms' % tot)
    print('%-6s %8s %8s %4s %4s %3s %5s %6s %9s %9s' % ('kind', 'n_in', 'n_out', 'ci', 'co', 'k', 'flip', 'calls', 'us/call', 'ms total'))
    for key, (c, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
        print('%-6s %8d %8d %4d %4d %3d %5d %6d %9.1f %9.3f' % (key + (c, ms / c * 1e3, ms)))


if __name__ == '__main__':
    main()
