"""The streamed weight gradient (csrc/wgrad_dma.hip wgrad_stream_kernel) against the offset-major one on the real
level-0 / level-1 kernel maps of a 5-scan batch, the stream tables built HERE in torch (the prototype of
lidal_wgrad_streams_build): time per launch, difference to the f64 gradient.
  LEVEL=0|1  CA=96 CB=96  BLOCK=4096  W=512  KEY=parent|row|none  REPS=20"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.network import SPVCNN, Geometry  # noqa: E402
from lidal_amd.nn.functional.conv import wgrad_scratch  # noqa: E402

UNIT = 65536


def slot_lengths(sizes, wx):
    """fixed-point slots per offset: proportional to the rule counts, at least one slot for an offset with rules
    (csrc/wgrad_streams.hip plan_kernel: clamped to the fixed point)"""
    k = len(sizes)
    total = sum(sizes)
    clamped = [False] * k
    free, rest = wx, total
    for _ in range(k):
        move = [(not c) and s > 0 and s * free < rest for s, c in zip(sizes, clamped)]
        clamped = [c or mv for c, mv in zip(clamped, move)]
        free = wx - sum(clamped)
        rest = total - sum(s for s, c in zip(sizes, clamped) if c)
        if not any(move):
            break
    return [0 if s == 0 else (UNIT if c else s * free * UNIT // max(rest, 1)) for s, c in zip(sizes, clamped)]


def build_streams(pairs, sizes, key, key_range, nb, w_total):
    """-> (spairs i32 [stages * 64, 2], sdesc i32, stats).  Rules sorted by (XCD = block % 8, offset, block); the list of
    an (XCD, offset) is cut into 32-rule steps, the steps are dealt to the workgroups that share the offset on that XCD
    (slots in proportion to the XCD's own rule counts, a low-discrepancy sequence spreads a workgroup's steps over the
    whole list), and a workgroup's steps are stored in block order."""
    dev = pairs.device
    k = len(sizes)
    wx = w_total // 8
    nbx = nb // 8
    koff = [0]
    for s in sizes:
        koff.append(koff[-1] + s)
    m = koff[-1]
    kk = torch.repeat_interleave(torch.arange(k, device=dev), torch.tensor(sizes, device=dev))
    out = pairs[:m, 1].long()
    kv = (key[out].long() if key is not None else out)
    b = (kv * nb) // key_range
    xcd = b % 8
    key1 = ((xcd * 32 + kk) * nbx) + b // 8
    order = torch.sort(key1, stable=True)[1]
    k1s = key1[order]
    cnt = torch.bincount(xcd * 32 + kk, minlength=256).view(8, 32)[:, :k]                   # rules of (xcd, offset)
    steps = (cnt + 31) // 32
    lstart = (torch.cumsum(cnt.reshape(-1), 0) - cnt.reshape(-1)).view(8, k)                  # first sorted position
    sbase = (torch.cumsum(steps.reshape(-1), 0) - steps.reshape(-1)).view(8, k)               # first step id
    t_total = int(steps.sum())
    cnt_h = cnt.tolist()
    wk = [[-1, -1] for _ in range(w_total)]
    kred = [[[0, 0, 0] for _ in range(8)] for _ in range(k)]
    starts_all, lens_all, first_all = [], [], []
    for x in range(8):
        lens = slot_lengths(cnt_h[x], wx)
        starts = [0] * k
        for i in range(1, k):
            starts[i] = starts[i - 1] + lens[i - 1]
        first = [-1] * wx
        for j in range(wx):
            for q in range(k):
                if lens[q] and starts[q] <= j * UNIT < starts[q] + lens[q]:
                    first[j] = q
                    wk[8 * j + x][0] = q
                if lens[q] and j * UNIT < starts[q] < (j + 1) * UNIT:
                    assert wk[8 * j + x][1] == -1
                    wk[8 * j + x][1] = q
        for q in range(k):
            if lens[q]:
                j0, j1 = starts[q] >> 16, (starts[q] + lens[q] - 1) >> 16
                kred[q][x] = [j0, j1 - j0 + 1, int(first[j0] != q)]
        starts_all.append(starts); lens_all.append(lens); first_all.append(first)
    starts_t = torch.tensor(starts_all, device=dev)
    lens_t = torch.tensor(lens_all, device=dev)
    first_t = torch.tensor(first_all, device=dev)
    # steps: (xcd, offset, s)
    xk = torch.repeat_interleave(torch.arange(8 * k, device=dev), steps.reshape(-1))
    sx, sk = xk // k, xk % k
    s_in = torch.arange(t_total, device=dev) - sbase.reshape(-1)[xk]
    first_rule = lstart.reshape(-1)[xk] + 32 * s_in
    blk = k1s[first_rule] % nbx
    u = (s_in * 2654435769) & 0xFFFFFFFF
    pos = starts_t[sx, sk] + ((u * lens_t[sx, sk]) >> 32)
    j = pos >> 16
    sset = (sk != first_t[sx, j]).long()
    w = 8 * j + sx
    key2 = (w * nbx + blk) * 2 + sset
    order2 = torch.sort(key2, stable=True)[1]
    per_w = torch.bincount(w, minlength=w_total)
    stages = (per_w + 1) // 2
    soff = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(stages, 0)])
    wfirst = torch.cumsum(per_w, 0) - per_w
    w_sorted = w[order2]
    dest_step = torch.empty(t_total, dtype=torch.long, device=dev)
    dest_step[order2] = soff[w_sorted] * 2 + (torch.arange(t_total, device=dev) - wfirst[w_sorted])
    # rules
    g = torch.arange(m, device=dev)
    xk_r = k1s // nbx
    r_in = g - lstart.reshape(-1)[(xk_r // 32) * k + xk_r % 32]
    t_r = sbase.reshape(-1)[(xk_r // 32) * k + xk_r % 32] + r_in // 32
    dst = dest_step[t_r] * 32 + r_in % 32
    total = int(soff[-1]) * 64
    spairs = torch.full((total, 2), 0x7FFFFFFF, dtype=torch.int32, device=dev)
    xv = pairs[:m, 0].long()[order] | (sset[t_r] << 31)
    xv = torch.where(xv >= 2 ** 31, xv - 2 ** 32, xv).int()
    spairs[dst, 0] = xv
    spairs[dst, 1] = pairs[:m, 1][order]
    sdesc = torch.cat([torch.tensor([w_total, k, total // 64, 0], device=dev), soff, torch.tensor(wk, device=dev).reshape(-1),
                       torch.tensor(kred, device=dev).reshape(-1)]).int()
    stats = dict(stages=total // 64, pad=total / m - 1, stages_min=int(stages.min()), stages_max=int(stages.max()))
    return spairs, sdesc, stats


def main():
    dev = torch.device('cuda', 0)
    level = int(os.environ.get('LEVEL', '0'))
    ca, cb = int(os.environ.get('CA', '96')), int(os.environ.get('CB', '96'))
    block = int(os.environ.get('BLOCK', '4096'))
    w_total = int(os.environ.get('W', '512'))
    keymode = os.environ.get('KEY', 'parent' if level == 0 else 'row')
    reps = int(os.environ.get('REPS', '20'))
    bt = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(bt['coords_v_b']).to(dev)
    model = SPVCNN(19).to(dev).train()
    g = Geometry.build(model, coords, True)
    s = 1 << level
    st = (s, s, s)
    km = g.x0.kmaps[(st, (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n = km.sizes[0]
    sizes = [int(v) for v in km.nbsizes.tolist()]
    pairs = km._nbmaps_cap
    key, key_range = None, n
    if keymode == 'parent':
        k2 = g.x0.kmaps[(st, (2, 2, 2), (2, 2, 2), (1, 1, 1))]
        p2 = k2._nbmaps_cap[:k2.total]
        key = torch.empty(n, dtype=torch.int32, device=dev)
        key[p2[:, 0].long()] = p2[:, 1]
        key_range = k2.sizes[1]
    elif keymode == 'none':          # a key without spatial meaning: what the blocks buy by themselves
        key = torch.randperm(n, device=dev).int()
    nb = max(8, (n + 8 * block - 1) // (8 * block) * 8)
    spairs, sdesc, stats = build_streams(pairs, sizes, key, key_range, nb, w_total)
    print('level', level, 'rows', n, 'rules', km.total, 'blocks', nb, 'W', w_total, 'key', keymode, stats)
    if os.environ.get('NATIVE', '1') != '0' and block == 1024 and keymode in ('parent', 'row'):
        # the library's builder against the prototype above, bit for bit; its time
        L = B.lib()
        cap = int(L.lidal_wgrad_streams_rules(n, 27, w_total))
        words = int(L.lidal_wgrad_streams_desc_words(27, w_total))
        ws_bytes = int(L.lidal_wgrad_streams_workspace_bytes(n, 27))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        sp2 = torch.full((cap, 2), -7, dtype=torch.int32, device=dev)
        sd2 = torch.full((words,), -7, dtype=torch.int32, device=dev)
        key_tab, key_k = (None, 0)
        if keymode == 'parent':
            key_tab, key_k = k2.nbr_in, 8

        def build():
            B.check(L.lidal_wgrad_streams_build(B.ptr(pairs), B.ptr(km.koff), 27, n, B.ptr(key_tab), key_k, key_range, w_total,
                                                B.ptr(sp2), cap, B.ptr(sd2), B.ptr(ws), ws_bytes, B.stream()), 'streams_build')
        build()
        torch.cuda.synchronize()
        total = spairs.shape[0]
        print('native builder: descriptor equal %s, rules equal %s (stages %d / %d), workspace %.0f MB' % (
            bool(torch.equal(sd2, sdesc)), bool(torch.equal(sp2[:total], spairs)), int(sd2[2]), total // 64, ws_bytes / 1e6))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            build()
        e1.record()
        torch.cuda.synchronize()
        print('native builder: %.1f us per build' % (e0.elapsed_time(e1) * 100))
        spairs, sdesc = sp2, sd2
    x = torch.randn(n, ca, device=dev).bfloat16()
    gy = torch.randn(n, cb, device=dev).bfloat16()
    L = B.lib()
    gw0 = torch.empty((27, ca, cb), dtype=torch.float32, device=dev)
    gw1 = torch.empty_like(gw0)
    gw2 = torch.empty_like(gw0)
    partial0 = wgrad_scratch(n, n, 27, ca, cb, torch.bfloat16, dev)
    partial1 = torch.empty((2 * w_total, ca, cb), dtype=torch.float32, device=dev)

    def old():
        B.check(L.lidal_conv_wgrad(B.ptr(x), B.ptr(gy), n, n, B.ptr(pairs), B.ptr(km.koff), 0, B.ptr(gw0), B.ptr(partial0),
                                   partial0.shape[0], 27, ca, cb, B.BF16, B.stream()), 'wgrad')

    def new(dst):
        B.check(L.lidal_conv_wgrad_streams(B.ptr(x), B.ptr(gy), n, n, B.ptr(spairs), B.ptr(sdesc), w_total, 0, B.ptr(dst),
                                           B.ptr(partial1), partial1.shape[0], 27, ca, cb, B.BF16, B.stream()), 'wgrad streams')

    old()
    new(gw1)
    partial1.fill_(float('nan'))
    new(gw2)
    torch.cuda.synchronize()
    ref = torch.zeros((27, ca, cb), dtype=torch.float64, device=dev)
    xd, gd = x.double(), gy.double()
    o = 0
    for kk, sz in enumerate(sizes):
        pr = pairs[o:o + sz].long()
        ref[kk] = xd[pr[:, 0]].t() @ gd[pr[:, 1]]
        o += sz
    scale = float(ref.abs().max())
    print('max |old - f64| / scale %.3g   max |streams - f64| / scale %.3g   streams bitwise twice: %s' % (
        float((gw0.double() - ref).abs().max()) / scale, float((gw1.double() - ref).abs().max()) / scale, bool(torch.equal(gw1, gw2))))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(2):
        ev[0].record()
        for _ in range(reps):
            old()
        ev[1].record()
        for _ in range(reps):
            new(gw1)
        ev[2].record()
        torch.cuda.synchronize()
    print('offset-major (+reduce) %.1f us   streams (+reduce) %.1f us' % (ev[0].elapsed_time(ev[1]) * 1e3 / reps,
                                                                         ev[1].elapsed_time(ev[2]) * 1e3 / reps))


if __name__ == '__main__':
    main()
