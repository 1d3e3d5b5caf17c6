"""TEST INFRASTRUCTURE -- a CPU restatement (numpy) of the stream tables of the weight gradient
(lidal_amd/csrc/wgrad_streams.hip lidal_wgrad_streams_build) and of what csrc/wgrad_dma.hip's wgrad_stream_kernel and
its reducer compute from them.  Only tests/ may import this.

This is NOT a restatement of the reference: torchsparse's convolution_backward_cuda (v1.4.0, the grad_weight loop:
gather the rules' rows, one GEMM per offset) fixes only the RESULT, gw[k] = sum over the rules (i, j) of offset k of
a[i]^T b[j] -- oracle/tsref restates that, and tests/test_teacher_forced_gpu.py holds the streamed launches to it.  The
tables are this library's own decomposition of that sum; the restatement below pins the device builder bit for bit
(tests/test_wgrad_streams_gpu.py) and is itself checked against the plain sum on the CPU (tests/test_streams_cpu.py).
"""
import numpy as np

UNIT = 65536            # fixed-point slot
MAXK = 32
MAXBX = 64
HDR = 4
BLOCK_ROWS = 1024
PAD = 0x7FFFFFFF


def blocks_for(n_rows):
    bx = min(max(-(-n_rows // (8 * BLOCK_ROWS)), 1), MAXBX)
    return 8 * bx


def slot_lengths(counts, wx):
    """Slots (1/65536 units) of the offsets on one XCD: proportional to the rule counts, at least one slot for an
    offset that has rules (plan_kernel: clamped to the fixed point, all offsets judged together per round)."""
    k = len(counts)
    total = int(sum(counts))
    clamped = [False] * k
    free, rest = wx, total
    for _ in range(k):
        move = [(not c) and s > 0 and s * free < rest for s, c in zip(counts, clamped)]
        clamped = [c or mv for c, mv in zip(clamped, move)]
        free = wx - sum(clamped)
        rest = total - sum(int(s) for s, c in zip(counts, clamped) if c)
        if not any(move):
            break
    return [0 if s == 0 else (UNIT if c else int(s) * free * UNIT // max(rest, 1)) for s, c in zip(counts, clamped)]


def build_streams(pairs, sizes, key, key_range, n_rows, n_wg):
    """pairs i32 [M, 2] grouped by offset (sizes [k]); key: None (the row index) or an int array [n_rows] in
    [0, key_range).  -> (spairs i32 [stages * 64, 2], sdesc i32 [HDR + (n_wg + 1) + 2 n_wg + 24 k])."""
    pairs = np.asarray(pairs)
    k = len(sizes)
    m = int(sum(sizes))
    wx = n_wg // 8
    nb = blocks_for(n_rows)
    kk = np.repeat(np.arange(k), sizes)
    out = pairs[:m, 1].astype(np.int64)
    kv = out if key is None else np.asarray(key).astype(np.int64)[out]
    b = np.minimum(kv * nb // key_range, nb - 1)
    xcd = b % 8
    key1 = (xcd * MAXK + kk) * MAXBX + b // 8
    order = np.argsort(key1, kind='stable')
    k1s = key1[order]
    cnt = np.bincount(xcd * MAXK + kk, minlength=8 * MAXK)                      # rules of (xcd, offset), sorted order
    steps = (cnt + 31) // 32
    lstart = np.cumsum(cnt) - cnt
    sbase = np.cumsum(steps) - steps
    t_total = int(steps.sum())
    starts = np.zeros((8, MAXK), np.int64)
    lens = np.zeros((8, MAXK), np.int64)
    first = np.full((8, 64), -1, np.int64)
    second = np.full((8, 64), -1, np.int64)
    for x in range(8):
        ln = slot_lengths([int(v) for v in cnt[x * MAXK:x * MAXK + k]], wx)
        lens[x, :k] = ln
        starts[x, :] = np.cumsum(lens[x]) - lens[x]
        for j in range(wx):
            for q in range(k):
                if lens[x, q] and starts[x, q] <= j * UNIT < starts[x, q] + lens[x, q]:
                    first[x, j] = q
                if lens[x, q] and j * UNIT < starts[x, q] < (j + 1) * UNIT:
                    assert second[x, j] == -1
                    second[x, j] = q
    # steps
    xk = np.repeat(np.arange(8 * MAXK), steps)
    sx, sk = xk // MAXK, xk % MAXK
    s_in = np.arange(t_total) - sbase[xk]
    blk = k1s[lstart[xk] + 32 * s_in] % MAXBX
    u = (s_in * 2654435769) & 0xFFFFFFFF
    pos = starts[sx, sk] + ((u * lens[sx, sk]) >> 32)
    j = pos >> 16
    sset = (sk != first[sx, j]).astype(np.int64)
    assert np.all((sset == 0) | (sk == second[sx, j]))
    w = 8 * j + sx
    key2 = (w * MAXBX + blk) * 2 + sset
    order2 = np.argsort(key2, kind='stable')
    per_w = np.bincount(w, minlength=n_wg)
    stages = (per_w + 1) // 2
    soff = np.concatenate([[0], np.cumsum(stages)])
    wfirst = np.cumsum(per_w) - per_w
    dest_step = np.empty(t_total, np.int64)
    ws = w[order2]
    dest_step[order2] = soff[ws] * 2 + (np.arange(t_total) - wfirst[ws])
    # rules
    xk_r = k1s // MAXBX
    r_in = np.arange(m) - lstart[xk_r]
    t_r = sbase[xk_r] + r_in // 32
    dst = dest_step[t_r] * 32 + r_in % 32
    total = int(soff[-1]) * 64
    sp = np.full((total, 2), PAD, np.int64)
    written = np.zeros(total, bool)
    sp[dst, 0] = pairs[:m, 0].astype(np.int64)[order] | (sset[t_r] << 31)
    sp[dst, 1] = pairs[:m, 1].astype(np.int64)[order]
    written[dst] = True
    # the padding of a list's partial last step carries the step's set
    for i in np.nonzero(cnt % 32)[0]:
        t = sbase[i] + steps[i] - 1
        at = dest_step[t] * 32
        sp[at + cnt[i] % 32:at + 32, 0] = PAD | (int(sset[t]) << 31)
    sp = np.where(sp >= 2 ** 31, sp - 2 ** 32, sp).astype(np.int32)
    wk = np.stack([first[np.arange(n_wg) % 8, np.arange(n_wg) // 8], second[np.arange(n_wg) % 8, np.arange(n_wg) // 8]], 1)
    kred = np.zeros((k, 8, 3), np.int64)
    for q in range(k):
        for x in range(8):
            if lens[x, q]:
                j0, j1 = starts[x, q] >> 16, (starts[x, q] + lens[x, q] - 1) >> 16
                kred[q, x] = (j0, j1 - j0 + 1, int(first[x, j0] != q))
    sdesc = np.concatenate([[n_wg, k, total // 64, 0], soff, wk.reshape(-1), kred.reshape(-1)]).astype(np.int32)
    return sp, sdesc


def run_streams(a, b, spairs, sdesc, a_col=0):
    """What wgrad_stream_kernel + wgrad_stream_reduce_kernel compute, in float64: every workgroup adds the products of
    its 32-rule steps into the accumulator set the step's first rule names, the reducer adds the slabs of an offset."""
    n_wg, k = int(sdesc[0]), int(sdesc[1])
    soff = sdesc[HDR:HDR + n_wg + 1]
    wk = sdesc[HDR + n_wg + 1:HDR + n_wg + 1 + 2 * n_wg].reshape(n_wg, 2)
    kred = sdesc[HDR + n_wg + 1 + 2 * n_wg:].reshape(k, 8, 3)
    ca, cb = a.shape[1], b.shape[1]
    slabs = np.zeros((2 * n_wg, ca, cb))
    sp = spairs.astype(np.int64) & 0xFFFFFFFF
    for w in range(n_wg):
        for h in range(2 * int(soff[w]), 2 * int(soff[w + 1])):
            r = sp[32 * h:32 * h + 32]
            flag = int(r[0, 0]) >> 31
            ia, ib = (r[:, 1], r[:, 0] & PAD) if a_col else (r[:, 0] & PAD, r[:, 1])
            ok = (ia < a.shape[0]) & (ib < b.shape[0])
            assert np.all((r[ok, 0] >> 31) == flag)
            slabs[2 * w + flag] += a[ia[ok]].T @ b[ib[ok]]
    gw = np.zeros((k, ca, cb))
    for q in range(k):
        for x in range(8):
            j0, nj, s0 = (int(v) for v in kred[q, x])
            for jj in range(nj):
                ww = 8 * (j0 + jj) + x
                st = s0 if jj == 0 else 0
                assert wk[ww, st] == q
                gw[q] += slabs[2 * ww + st]
    return gw
