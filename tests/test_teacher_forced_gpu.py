"""bf16 parity INSIDE the whole step, where nothing amplifies (round 4).

End to end, a bf16 train step of this randomly initialised 49-layer net cannot be compared with ANY reference at
gradient level: two runs of the bf16-emulating oracle that differ only in accumulation arithmetic already disagree
at cosine 0.93-0.99 (tests/test_benchsize_gpu.py::test_bf16_train_step_matches_the_bf16_emulating_oracle).  So the
step is checked TEACHER-FORCED: one planned bf16 train step (network/plan.py) runs on the GPU with tracing on, and
then EVERY operation of its two launch plans -- each convolution / data gradient / weight gradient, each BatchNorm
forward and backward, every voxel <-> point exchange, mask, sum, concatenation, cast -- is replayed on the CPU by the
oracle's operator (oracle.tsref's torchsparse restatement, torch's batch_norm + autograd, in float64) ON THE
OPERATION'S OWN STORED OPERANDS, read back from the addresses in the plan.  Each stored output must equal the
float64 result rounded once to bf16, up to the rounding flips an f32 accumulation causes:
  * bf16 outputs: |stored - f64| <= 1 bf16 ulp of the value (+ 2^-17 of the tensor's largest magnitude for sums
    that cancel) on EVERY element, and stored == round(f64) on all but a small fraction of the elements;
  * f32 outputs (weight gradients, BatchNorm statistics and parameter gradients): relative error <= 1e-4 / 1e-5.
The operands of an operation are the HIP path's own outputs of the operations before it, so an error cannot hide
behind -- or be blamed on -- anything upstream.  Both networks, at 20 k points (every operation) and at the
benchmarked ~120 k points (every third convolution-type operation, everything else in full)."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


# ---- reading the plan's memory -----------------------------------------------------------------------------
def _peek(addr, nbytes):
    from lidal_amd import backend as B
    buf = np.empty(max(int(nbytes), 1), dtype=np.uint8)
    B.check(B.lib().lidal_debug_read(ctypes.c_void_p(int(addr)), buf.ctypes.data_as(ctypes.c_void_p), int(nbytes)),
            'debug_read')
    return buf[:int(nbytes)]


def _mat(addr, rows, cols, code, stride=None):
    """[rows, cols] matrix of dtype code (0 f32, 1 bf16) at `addr`, rows `stride` elements apart -> float64 tensor."""
    stride = cols if stride is None else int(stride)
    esz = 2 if code == 1 else 4
    n = (rows - 1) * stride + cols if rows > 0 else 0
    raw = _peek(addr, n * esz)
    if code == 1:
        t = torch.from_numpy(raw.view(np.int16).copy()).view(torch.bfloat16)
    else:
        t = torch.from_numpy(raw.view(np.float32).copy())
    if stride != cols:
        t = torch.as_strided(t, (rows, cols), (stride, 1))
    return t.reshape(rows, cols).double() if stride == cols else t.double()


def _vec(addr, n, dtype=np.float32):
    return torch.from_numpy(_peek(addr, n * np.dtype(dtype).itemsize).view(dtype).copy())


def _as_double(word):
    return float(np.array([word], dtype=np.int64).view(np.float64)[0])


def _ulp(x):
    """One bf16 ulp at |x| (float64 tensor)."""
    e = torch.floor(torch.log2(x.abs().clamp_min(2.0 ** -126)))
    return torch.pow(2.0, e - 7)


class _Stats:
    def __init__(self):
        self.rows = {}

    def bf16(self, kind, stored, ref, what, first=None):
        """`first`: the operation is a fused sum round(round(first) + other) -- the kernels round the summand they
        produce before adding, as the separate operators would -- so a flipped rounding of `first` moves the result
        by one ulp OF `first`, which may be many ulps of a sum that cancels."""
        r = ref.to(torch.bfloat16).double()
        scale = float(ref.abs().max())
        err = (stored - ref).abs()
        tol = _ulp(ref) + scale * 2.0 ** -17
        if first is not None:
            tol = tol + _ulp(first)
        worst = float((err / tol).max()) if err.numel() else 0.0
        flips = float((stored != r).double().mean()) if err.numel() else 0.0
        row = self.rows.setdefault(kind, [0, 0.0, 0.0, ''])
        row[0] += 1
        if worst > row[1]:
            row[1], row[3] = worst, what
        row[2] = max(row[2], flips)
        assert worst <= 1.0, (kind, what, 'an element is more than one bf16 ulp from the float64 result', worst)
        assert flips <= 0.02, (kind, what, 'too many elements differ from round(float64 result)', flips)

    def f32(self, kind, got, ref, what, bar):
        rel = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        row = self.rows.setdefault(kind, [0, 0.0, 0.0, ''])
        row[0] += 1
        if rel > row[1]:
            row[1], row[3] = rel, what
        assert rel <= bar, (kind, what, rel, bar)

    def exact(self, kind, got, ref, what):
        row = self.rows.setdefault(kind, [0, 0.0, 0.0, ''])
        row[0] += 1
        assert torch.equal(got, ref), (kind, what)

    def show(self, title):
        print(title)
        for k, (n, worst, flips, what) in sorted(self.rows.items()):
            print('   %-26s ops %4d   worst %.3g   largest fraction of flipped roundings %.2e   (%s)' % (k, n, worst, flips, what))


# ---- the replay ----------------------------------------------------------------------------------------------
def _replay(run, model, stats, conv_every=1):
    from lidal_amd import backend as B
    from lidal_amd.network import plan as P
    from oracle.tsref.nn.functional import _conv_apply, spdevoxelize, spvoxelize
    L = B.lib()
    prog, g = run.prog, run.geometry
    bf = run.code == 1
    names = {id(p): k for k, p in model.named_parameters()}
    pname = [names[id(p)] for p in prog.params]
    by_ptr = {p.data_ptr(): p for p in prog.params}
    for t in prog.buffers:
        by_ptr[t.data_ptr()] = t
    tables, rules = {}, {}
    for km in g.x0.kmaps.values():
        tables[km.order_out.table.data_ptr()] = (km, False)
        if km._order_in is not None:
            tables[km._order_in.table.data_ptr()] = (km, True)
        rules[km._nbmaps_cap.data_ptr()] = km
        if km._streams is not None:         # the same rules as one stream per workgroup (OP_CONV_WGRAD_STREAMS)
            rules[km._streams[0].data_ptr()] = km
    images = {}
    for c in prog.convs:
        for pf, pb in c.ptrs.values():
            images[pf], images[pb] = (c, False), (c, True)
    flat0 = run.flat_t.data_ptr()
    slot_of = {flat0 + 4 * prog.slot[i]: i for i in range(len(prog.params))}
    cpu_rules = {}

    def rules_of(km):
        if id(km) not in cpu_rules:
            cpu_rules[id(km)] = (km.nbmaps.cpu().long(), km.nbsizes.cpu().long(), tuple(km.sizes))
        return cpu_rules[id(km)]

    def operand(c, backward, n_red, n_col):
        """The [k, n_red, n_col] float64 operand an image of layer `c` stands for (bf16-rounded in bf16 mode), zero
        padded to the channel counts the launch was given."""
        w = c.param.detach().cpu()
        m = w.t().reshape(1, c.ci, c.co) if c.role == 1 else w.reshape(c.k, c.ci, c.co)
        if bf:
            m = m.to(torch.bfloat16)
        m = m.double()
        if backward:
            m = m.transpose(1, 2)
        out = torch.zeros(m.shape[0], n_red, n_col, dtype=torch.float64)
        out[:, :m.shape[1], :m.shape[2]] = m
        return out

    bn_eps = {float(m.eps) for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d)}
    assert len(bn_eps) == 1
    bn_eps = bn_eps.pop()
    torch.cuda.synchronize()
    n_conv = 0
    pending = None              # a weight gradient that went to scratch: checked through the copy / transpose after it
    for phase, words in run.tapes:
        i = 0
        while i < len(words):
            kind = words[i] & 0xFFFF
            na = L.lidal_plan_op_args(kind)
            a = words[i + 1:i + 1 + na]
            i += 1 + na
            if kind in (P.OP_CONV_APPLY_IMAGE, P.OP_CONV_DGRAD_BN_SUMS, P.OP_CONV_APPLY_IMAGE_WS, P.OP_CONV_DGRAD_BN_SUMS_WS):
                x_p, img, tab, _, _, out_p, n_in, n_out, ci, co, k, kflip, code = a[:13]
                n_conv += 1
                if n_conv % conv_every:
                    continue
                c, backward = images[img]
                x = _mat(x_p, n_in, ci, code)
                w = operand(c, backward, ci, co)
                if tab == 0:
                    ref = x @ w[0]
                    what = '%s dense %s %d->%d' % (pname[c.w], 'dgrad' if backward else 'fwd', ci, co)
                else:
                    km, inv = tables[tab]
                    nbmaps, nbsizes, sizes = rules_of(km)
                    ref = _conv_apply(x, w, nbmaps, nbsizes, sizes, bool(inv) != bool(kflip))
                    what = '%s k%d %s %d->%d rows %d' % (pname[c.w], k, 'dgrad' if backward else 'fwd', ci, co, n_out)
                assert ref.shape == (n_out, co), (what, ref.shape)
                if kind in (P.OP_CONV_APPLY_IMAGE, P.OP_CONV_APPLY_IMAGE_WS):
                    scale, shift, relu, res = a[13:17]
                    if shift:
                        ref = ref * _vec(scale, co).double() + _vec(shift, co).double()
                    assert relu == 0
                    first = None
                    if res:         # the kernel rounds its own product first, then adds (as the separate sum would)
                        first = ref
                        ref = (ref.to(torch.bfloat16).double() if code == 1 else ref) + _mat(res, n_out, co, code)
                    st = a[17]
                    if st:          # (count, mean, M2) per 128-row tile of the kernel's row order: merged, the batch statistics
                        tiles = -(-n_out // run.tile)
                        ts_ = _mat(st, tiles * co, 3, 0).reshape(co, tiles, 3).permute(1, 0, 2)      # stored [co][tiles][3]
                        stored = _mat(out_p, n_out, co, code)
                        cnt = ts_[:, :, 0].sum(0)
                        mean = (ts_[:, :, 0] * ts_[:, :, 1]).sum(0) / cnt
                        m2 = (ts_[:, :, 2] + ts_[:, :, 0] * (ts_[:, :, 1] - mean) ** 2).sum(0)
                        assert torch.equal(cnt, torch.full((co,), float(n_out), dtype=torch.float64)), what
                        stats.f32('conv tile statistics', mean, stored.mean(0), what + ' mean', 1e-4)
                        stats.f32('conv tile statistics', m2 / n_out, stored.var(0, unbiased=False), what + ' var', 1e-4)
                else:
                    first = None
                    bx, mean_p, inv_p, gam, bet, brelu, sums_p = a[13:20]
                    stored = _mat(out_p, n_out, co, code)
                    xb = _mat(bx, n_out, co, code)
                    mu, istd = _vec(mean_p, co).double(), _vec(inv_p, co).double()
                    xh = (xb - mu) * istd
                    dy = stored.clone()
                    if brelu:
                        dy[~(xh * _vec(gam, co).double() + _vec(bet, co).double() > 0)] = 0
                    tiles = -(-n_out // run.tile)
                    sm = _mat(sums_p, tiles * co, 2, 0).reshape(co, tiles, 2).sum(1)           # stored [co][tiles][2]
                    stats.f32('dgrad tile sums', sm[:, 0], dy.sum(0), what + ' sum dy', 1e-4)
                    stats.f32('dgrad tile sums', sm[:, 1], (dy * xh).sum(0), what + ' sum dy xhat', 1e-4)
                stored = _mat(out_p, n_out, co, code)
                if code == 1:
                    stats.bf16('conv ' + ('dgrad' if backward else 'fwd') + (' dense' if tab == 0 else '')
                               + (' + residual' if first is not None else ''), stored, ref, what, first)
                else:
                    stats.f32('conv f32', stored, ref, what, 1e-5)
            elif kind in (P.OP_CONV_WGRAD, P.OP_CONV_WGRAD_STREAMS):
                # (the streamed form names its rules by the stream table: the SAME rule lists, summed in another order --
                #  replayed from the map's own lists like the offset-major form)
                if kind == P.OP_CONV_WGRAD_STREAMS:
                    a_p, b_p, n_a, n_b, pairs, _, _, a_col, gw_p, _, _, k, ca, cb, code = a
                    assert pairs in rules and rules[pairs]._streams[0].data_ptr() == pairs
                else:
                    a_p, b_p, n_a, n_b, pairs, koff, a_col, gw_p, _, _, k, ca, cb, code = a
                n_conv += 1
                if n_conv % conv_every:
                    pending = None
                    continue
                xa, xb = _mat(a_p, n_a, ca, code), _mat(b_p, n_b, cb, code)
                ref = torch.zeros(k, ca, cb, dtype=torch.float64)
                if pairs == 0:
                    ref[0] = xa.t() @ xb
                else:
                    nbmaps, nbsizes, _ = rules_of(rules[pairs])
                    cur = 0
                    for kk in range(k):
                        m = int(nbsizes[kk])
                        pr = nbmaps[cur:cur + m]
                        cur += m
                        if m:
                            ia, ib = (pr[:, 1], pr[:, 0]) if a_col else (pr[:, 0], pr[:, 1])
                            ref[kk] = xa[ia].t() @ xb[ib]
                if gw_p in slot_of:
                    got = _mat(gw_p, k * ca, cb, 0).reshape(k, ca, cb)
                    stats.f32('weight gradient' + (' (streams)' if kind == P.OP_CONV_WGRAD_STREAMS else ''), got, ref,
                              pname[slot_of[gw_p]], 1e-4)
                    pending = None
                else:
                    pending = (gw_p, ref)
            elif kind == P.OP_TRANSPOSE_F32:
                src, sstride, dst, rows, cols = a
                if pending is not None and pending[0] == src:
                    ref = pending[1][0][:, :cols].t().contiguous()
                    stats.f32('weight gradient', _mat(dst, cols, rows, 0), ref, pname[slot_of[dst]] + ' (transposed)', 1e-4)
                pending = None
            elif kind == P.OP_COPY2D:
                src, spitch, dst, dpitch, rows, rbytes, zbytes = a
                if pending is not None and pending[0] == src and dst in slot_of:        # the channel-padded stem
                    k_, ca_, cb_ = pending[1].shape
                    keep = rbytes // 4 // cb_
                    stats.f32('weight gradient', _mat(dst, rows * keep, cb_, 0).reshape(rows, keep, cb_),
                              pending[1][:, :keep], pname[slot_of[dst]] + ' (padded input)', 1e-4)
                    pending = None
                    continue
                sb = torch.from_numpy(_peek(src, (rows - 1) * spitch + rbytes).copy())
                db = torch.from_numpy(_peek(dst, (rows - 1) * dpitch + rbytes + zbytes).copy())
                s2 = torch.as_strided(sb, (rows, rbytes), (spitch, 1))
                d2 = torch.as_strided(db, (rows, rbytes + zbytes), (dpitch, 1))
                stats.exact('copy2d', d2[:, :rbytes], s2, 'copy')
                assert not d2[:, rbytes:].any()
            elif kind in (P.OP_BN_TRAIN_FWD, P.OP_BN_TRAIN_FWD_TILES):
                x_p, code, n, c, gam, bet, eps, _, _, _, _, relu, res, y_p, mean_p, inv_p = a[:16]
                x = _mat(x_p, n, c, code)
                gamma, beta = _vec(gam, c).double(), _vec(bet, c).double()
                mu, var = x.mean(0), x.var(0, unbiased=False)
                istd = 1.0 / torch.sqrt(var + _as_double(eps))
                stats.f32('bn statistics', _vec(mean_p, c).double(), mu, 'mean %d x %d' % (n, c), 1e-5)
                stats.f32('bn statistics', _vec(inv_p, c).double(), istd, 'invstd %d x %d' % (n, c), 1e-5)
                y = (x - mu) * istd * gamma + beta
                if relu & 1:
                    y = y.clamp_min(0)
                first = None
                if res:
                    first = y
                    y = (y.to(torch.bfloat16).double() if code == 1 else y) + _mat(res, n, c, code)
                    if relu & 2:
                        y = y.clamp_min(0)
                if code == 1:
                    stats.bf16('bn forward' + (' + residual' if res else ''), _mat(y_p, n, c, code), y, '%d x %d relu %d' % (n, c, relu), first)
                else:
                    stats.f32('bn forward f32', _mat(y_p, n, c, code), y, '%d x %d' % (n, c), 1e-5)
            elif kind in (P.OP_BN_BWD, P.OP_BN_BWD_TILES, P.OP_BN_BWD_FROM_SUMS):
                x_p, dy_p, ldy, code, n, c, gam, bet, relu, mean_p, inv_p, dx_p, gg_p, gb_p = a[:14]
                x = _mat(x_p, n, c, code).requires_grad_(True)
                dy = _mat(dy_p, n, c, code, ldy)
                gamma = _vec(gam, c).double().requires_grad_(True)
                beta = _vec(bet, c).double().requires_grad_(True)
                mu, var = x.mean(0), x.var(0, unbiased=False)
                y = (x - mu) / torch.sqrt(var + bn_eps) * gamma + beta
                border = None
                if relu:            # a value within f32 rounding of zero may sit on the other side of the ReLU in the kernel
                    border = y.detach().abs() < 2e-6 * y.detach().abs().max()
                    y = y.clamp_min(0)
                gx, gg, gb = torch.autograd.grad(y, (x, gamma, beta), dy)
                got = _mat(dx_p, n, c, code)
                if border is not None and bool(border.any()):
                    assert float(border.double().mean()) < 1e-3
                    got = torch.where(border, gx, got)
                if code == 1:
                    stats.bf16('bn backward dx', got, gx, '%d x %d' % (n, c))
                else:
                    stats.f32('bn backward dx f32', got, gx, '%d x %d' % (n, c), 1e-5)
                # (f32 sums over a tile / a slab of rows, merged in f64: the convolutions' tile sums, and since round 5 every
                #  bf16 lidal_bn_bwd -- csrc/bn.hip bn_bwd_slab_sums_kernel; the f32 mode keeps f64 sums and the 1e-5 bar)
                tile_bar = kind == P.OP_BN_BWD_TILES or (kind == P.OP_BN_BWD and code == 1)
                stats.f32('bn backward grad gamma', _vec(gg_p, c).double(), gg, '%d x %d' % (n, c), 2e-4 if tile_bar else 1e-5)
                stats.f32('bn backward grad beta', _vec(gb_p, c).double(), gb, '%d x %d' % (n, c), 2e-4 if tile_bar else 1e-5)
            elif kind in (P.OP_ADD_RELU_BWD_BN_SUMS, P.OP_ADD_RELU_BWD_BN_TILE_SUMS):      # (the sums they leave are checked through OP_BN_BWD_FROM_SUMS / OP_BN_BWD_TILES)
                y_p, g_p, gin_p, code, n, c = a[:6]
                y, gg = _mat(y_p, n, c, code), _mat(g_p, n, c, code)
                stats.exact('relu mask', _mat(gin_p, n, c, code), torch.where(y > 0, gg, torch.zeros_like(gg)), 'mask')
            elif kind == P.OP_ADD_RELU_BWD:
                y_p, g_p, gin_p, numel, code = a
                y, gg = _mat(y_p, 1, numel, code), _mat(g_p, 1, numel, code)
                stats.exact('relu mask', _mat(gin_p, 1, numel, code), torch.where(y > 0, gg, torch.zeros_like(gg)), 'mask')
            elif kind == P.OP_ADD2D:
                a_p, sa, b_p, sb_, o_p, so, rows, c, code = a
                ref = _mat(a_p, rows, c, code, sa) + _mat(b_p, rows, c, code, sb_)
                got = _mat(o_p, rows, c, code, so)
                stats.exact('sum of two gradients', got, ref.to(torch.bfloat16).double() if code == 1 else ref.float().double(), 'add')
            elif kind == P.OP_CAST_ROWS_BF16:
                src, c_src, dst, c_dst, n = a
                x = _mat(src, n, c_src, 0)
                ref = torch.zeros(n, c_dst, dtype=torch.float64)
                ref[:, :c_src] = x.to(torch.bfloat16).double()
                stats.exact('cast', _mat(dst, n, c_dst, 1), ref, 'cast')
            elif kind == P.OP_VOXELIZE_FWD_1TO1:
                f_p, idx_p, o_p, n, c, code = a
                idx = _vec(idx_p, n, np.int32).long()
                ref = torch.zeros(n, c, dtype=torch.float64)
                ref[idx] = _mat(f_p, n, c, code)
                stats.exact('voxelize 1:1', _mat(o_p, n, c, code), ref, 'rows')
            elif kind == P.OP_VOXELIZE_FWD_SORTED:
                f_p, ord_p, seg_p, cnt_p, o_p, m, c, code, n_ent = a[:9]
                order = _vec(ord_p, n_ent, np.int32).long()
                seg = _vec(seg_p, m + 1, np.int64)
                idx = torch.full((n_ent,), -1, dtype=torch.long)
                owner = torch.repeat_interleave(torch.arange(m), seg[1:] - seg[:-1])
                idx[order[:owner.numel()]] = owner
                ref = spvoxelize(_mat(f_p, n_ent, c, code), idx, _vec(cnt_p, m, np.int32))
                stats.bf16('voxelize', _mat(o_p, m, c, code), ref, '%d -> %d x %d' % (n_ent, m, c)) if code == 1 else \
                    stats.f32('voxelize f32', _mat(o_p, m, c, code), ref, 'rows', 1e-5)
            elif kind == P.OP_VOXELIZE_BWD:
                g_p, idx_p, cnt_p, res, gin_p, n, m, c, code = a
                idx = _vec(idx_p, n, np.int32).long()
                cnt = _vec(cnt_p, m, np.int32).double()
                ref = _mat(g_p, m, c, code)[idx] / cnt[idx].unsqueeze(1)
                first = None
                if res:
                    first = ref
                    ref = (ref.to(torch.bfloat16).double() if code == 1 else ref) + _mat(res, n, c, code)
                stats.bf16('voxelize backward', _mat(gin_p, n, c, code), ref, '%d x %d' % (n, c), first) if code == 1 else \
                    stats.f32('voxelize backward f32', _mat(gin_p, n, c, code), ref, 'rows', 1e-5)
            elif kind == P.OP_DEVOXELIZE_FWD:
                f_p, idx_p, w_p, o_p, n, m, c, code = a
                idx8 = _vec(idx_p, n * 8, np.int32).reshape(n, 8)
                w8 = _vec(w_p, n * 8).reshape(n, 8).double()
                ref = spdevoxelize(_mat(f_p, m, c, code), idx8, w8)
                stats.bf16('devoxelize', _mat(o_p, n, c, code), ref, '%d -> %d x %d' % (m, n, c)) if code == 1 else \
                    stats.f32('devoxelize f32', _mat(o_p, n, c, code), ref, 'rows', 1e-5)
            elif kind == P.OP_DEVOXELIZE_BWD_SORTED:
                g_p, ord_p, seg_p, w_p, gin_p, m, c, code, n_ent = a[:9]
                order = _vec(ord_p, n_ent, np.int32).long()
                seg = _vec(seg_p, m + 1, np.int64)
                w = _vec(w_p, n_ent).double()
                gout = _mat(g_p, n_ent // 8, c, code)
                owner = torch.repeat_interleave(torch.arange(m), seg[1:] - seg[:-1])
                ent = order[:owner.numel()]
                ref = torch.zeros(m, c, dtype=torch.float64).index_add(0, owner, gout[ent // 8] * w[ent].unsqueeze(1))
                stats.bf16('devoxelize backward', _mat(gin_p, m, c, code), ref, '%d x %d' % (m, c)) if code == 1 else \
                    stats.f32('devoxelize backward f32', _mat(gin_p, m, c, code), ref, 'rows', 1e-5)
            elif kind == P.OP_DEVOXELIZE_BWD_CELLS:
                # (gout, vorder, vseg, w8, corder, cseg, gin, m, c, dtype, ws, bytes): the reference is the definition --
                # gin[v] = sum over points p and corners j with corner j of p's cell == v of w8[p][j] * gout[p] -- with the
                # cells' corner indices read back out of the voxels' lists
                g_p, vo_p, vs_p, w_p, co_p, cs_p, gin_p, m, c, code = a[:10]
                vseg = _vec(vs_p, m + 1, np.int64)
                n_pts = int(vseg[-1])
                vorder = _vec(vo_p, n_pts, np.int32).long()
                cseg = _vec(cs_p, m + 1, np.int64)
                corder = _vec(co_p, int(cseg[-1]), np.int32).long()
                cidx = torch.full((m * 8,), -1, dtype=torch.long)
                cidx[corder] = torch.repeat_interleave(torch.arange(m), cseg[1:] - cseg[:-1])
                cidx = cidx.view(m, 8)
                n_all = run.T.p
                cell_of = torch.full((n_all,), -1, dtype=torch.long)
                cell_of[vorder] = torch.repeat_interleave(torch.arange(m), vseg[1:] - vseg[:-1])
                assert bool((cell_of >= 0).all())
                w = _vec(w_p, n_all * 8).double().view(n_all, 8)
                gout = _mat(g_p, n_all, c, code)
                ref = torch.zeros(m, c, dtype=torch.float64)
                for j in range(8):
                    tgt = cidx[cell_of, j]
                    ok = tgt >= 0
                    ref.index_add_(0, tgt[ok], gout[ok] * w[ok, j].unsqueeze(1))
                stats.bf16('devoxelize backward (cells)', _mat(gin_p, m, c, code), ref, '%d x %d' % (m, c)) if code == 1 else \
                    stats.f32('devoxelize backward (cells) f32', _mat(gin_p, m, c, code), ref, 'rows', 1e-5)
            elif kind == P.OP_COLSUM:
                x_p, code, n, c, o_p = a[:5]
                keep = c
                if o_p in slot_of:
                    keep = min(c, prog.params[slot_of[o_p]].numel())
                stats.f32('bias gradient', _vec(o_p, keep).double(), _mat(x_p, n, c, code).sum(0)[:keep], 'colsum %d x %d' % (n, c), 1e-5)
            elif kind in (P.OP_FORK_SIDE, P.OP_JOIN_SIDE):
                pass                            # (stream ordering only)
            else:
                raise AssertionError('operation kind %d of the plan is not replayed' % kind)


def _step(name, points, autocast, conv_every):
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet, plan
    from lidal_amd.train_step import forward_backward
    from weights import fill_state_dict
    b = synth.make_train_batch(n_frames=1, n_points=points, seed=7122)
    coords, feats, labels = (torch.from_numpy(b[k]).to(DEV) for k in ('coords_v_b', 'feats_v_b', 'labels_v_b'))
    model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).to(DEV).train()
    if hasattr(model, 'dropout'):
        model.dropout.p = 0.0
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    saved = plan.TRACE, plan.ENABLED
    plan.TRACE, plan.ENABLED = [], True
    try:
        loss, _ = forward_backward(model, feats, coords, labels, autocast=autocast)
        assert np.isfinite(loss.item()) and len(plan.TRACE) == 1
        run = plan.TRACE[0]
    finally:
        plan.TRACE, plan.ENABLED = saved
    stats = _Stats()
    try:
        _replay(run, model, stats, conv_every)
        # the gradients autograd delivered are the flat buffer's slots
        for i, p in enumerate(run.prog.params):
            assert p.grad is not None and p.grad.data_ptr() == run.flat_t.data_ptr() + 4 * run.prog.slot[i]
    finally:
        run.release()
    stats.show('%s, %d voxels, %s: every operation of the planned step replayed by the oracle on its own operands'
               % (name, coords.shape[0], 'bf16' if autocast else 'f32'))
    return stats


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_every_operation_of_a_bf16_step_against_the_oracle_on_its_own_operands(name):
    stats = _step(name, 20000, True, 1)
    assert sum(r[0] for r in stats.rows.values()) > 300


def test_streamed_weight_gradients_teacher_forced(monkeypatch):
    """Round 6: the weight gradients of the large levels run on rule streams (plan op 34, lidal_conv_wgrad_streams).  With
    the row threshold forced down to test size every streamed launch of a step is replayed like the others."""
    from lidal_amd import backend as B
    monkeypatch.setattr(B, 'WGRAD_STREAMS_ROWS', 3000)
    stats = _step('spvcnn', 20000, True, 1)
    assert stats.rows['weight gradient (streams)'][0] >= 6 and stats.rows['weight gradient'][0] >= 20, stats.rows


def test_every_operation_of_an_f32_step_against_the_oracle_on_its_own_operands():
    _step('spvcnn', 12000, False, 1)


@pytest.mark.parametrize('name', ['spvcnn', 'minkunet'])
def test_the_benchmarked_scan_teacher_forced(name):
    """BASELINE.json's scan (~120 k points, ~83 k voxels): every third convolution-type operation (forward, data
    gradient, weight gradient -- 145 of them), everything else in full."""
    _step(name, 120000, True, 3)


# ---- the inference pass (round 5): every convolution / dense layer of ONE planned 8-view pass, f32 in the split form ----
def _replay_inference(run, model, stats):
    """The conv-type operations of an inference plan (network/plan.py _EvalRun: Conv3d / Linear with the folded BatchNorm
    map, ReLU and residual sum in the epilogue) replayed in float64 on their own stored operands."""
    from lidal_amd import backend as B
    from lidal_amd.network import plan as P
    from oracle.tsref.nn.functional import _conv_apply
    L = B.lib()
    prog, g = run.prog, run.geometry
    names = {id(p): k for k, p in model.named_parameters()}
    pname = [names[id(p)] for p in prog.params]
    tables = {}
    for km in g.x0.kmaps.values():
        tables[km.order_out.table.data_ptr()] = (km, False)
        if km._order_in is not None:
            tables[km._order_in.table.data_ptr()] = (km, True)
    images = {}
    for c in prog.convs:
        for pf, _ in c.ptrs.values():
            images[pf] = c
    cpu_rules = {}
    codes = {}
    torch.cuda.synchronize()
    for phase, words in run.tapes:
        i = 0
        while i < len(words):
            kind = words[i] & 0xFFFF
            na = L.lidal_plan_op_args(kind)
            a = words[i + 1:i + 1 + na]
            i += 1 + na
            if kind not in (P.OP_CONV_APPLY_IMAGE, P.OP_CONV_APPLY_IMAGE_WS):
                continue
            x_p, img, tab, _, _, out_p, n_in, n_out, ci, co, k, kflip, code, scale, shift, relu, res = a[:17]
            c = images[img]
            codes[code] = codes.get(code, 0) + 1
            data_code = 1 if code == 1 else 0                       # (LIDAL_F32_SPLIT: f32 rows in and out)
            x = _mat(x_p, n_in, ci, data_code)
            w = c.param.detach().cpu()
            w = (w.t().reshape(1, c.ci, c.co) if c.role == 1 else w.reshape(c.k, c.ci, c.co))
            w = (w.to(torch.bfloat16) if code == 1 else w).double()
            wp = torch.zeros(w.shape[0], ci, co, dtype=torch.float64)
            wp[:, :w.shape[1], :w.shape[2]] = w
            if tab == 0:
                ref = x @ wp[0]
                what = '%s dense %d->%d rows %d' % (pname[c.w], ci, co, n_out)
            else:
                km, inv = tables[tab]
                if id(km) not in cpu_rules:
                    cpu_rules[id(km)] = (km.nbmaps.cpu().long(), km.nbsizes.cpu().long(), tuple(km.sizes))
                ref = _conv_apply(x, wp, *cpu_rules[id(km)], bool(inv) != bool(kflip))
                what = '%s k%d %d->%d rows %d' % (pname[c.w], k, ci, co, n_out)
            if shift:
                ref = ref * (_vec(scale, co).double() if scale else 1.0) + _vec(shift, co).double()
            if relu & 1:
                ref = ref.clamp_min(0)
            first = None
            if res:
                first = ref
                ref = (ref.to(torch.bfloat16).double() if code == 1 else ref) + _mat(res, n_out, co, data_code)
                if relu & 2:
                    ref = ref.clamp_min(0)
            stored = _mat(out_p, n_out, co, data_code)
            if code == 1:
                stats.bf16('inference conv' + (' + residual' if first is not None else ''), stored, ref, what, first)
            else:
                stats.f32('inference conv f32 %s' % ('split' if code == 2 else 'exact'), stored, ref, what, 2e-5)
    return codes


@pytest.mark.parametrize('name,autocast', [('spvcnn', False), ('minkunet', False), ('spvcnn', True)])
def test_every_convolution_of_an_inference_pass_against_the_oracle_on_its_own_operands(name, autocast):
    """One planned 8-view inference pass (score/prob_inference.py:97-99) teacher-forced: every Conv3d / Linear launch --
    in f32 the SPLIT form (three bf16 pieces per operand, six partial products on the bf16 matrix cores), with the folded
    BatchNorm map, ReLU and residual sum of its epilogue -- against the float64 product of its own stored operands:
    2e-5 of the output scale in f32 (the exact-f32 kernel's bar), half a bf16 ulp under bf16."""
    from lidal_amd import backend as B
    from lidal_amd import synth
    from lidal_amd.network import SPVCNN, MinkUNet, plan
    from lidal_amd.score.prob_inference import infer_frame
    from weights import fill_state_dict
    model = fill_state_dict({'spvcnn': SPVCNN, 'minkunet': MinkUNet}[name](19)).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    for m in model.modules():               # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.5)
    seq = synth.make_sequence(1, n_points=6000, seed=31)[0]
    sb = synth.make_score_batch(seq['points'], seq['intensity'], np.random.default_rng(2), inf_reps=8)
    c, f, inv = (torch.from_numpy(sb[k]).to(DEV) for k in ('coords_v_b', 'feats_v_b', 'inverse_indices_b'))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    saved = plan.TRACE, plan.ENABLED
    plan.TRACE, plan.ENABLED = [], True
    try:
        prob, pred = infer_frame(model, c, f, inv, 8, autocast=autocast)
        assert torch.isfinite(prob).all() and len(plan.TRACE) == 1
        run = plan.TRACE[0]
    finally:
        plan.TRACE, plan.ENABLED = saved
    stats = _Stats()
    try:
        codes = _replay_inference(run, model, stats)
    finally:
        run.release()
    stats.show('%s inference, %d voxels (8 views), %s: every convolution of the planned pass replayed on its own operands'
               % (name, c.shape[0], 'bf16' if autocast else 'f32'))
    if autocast:
        assert set(codes) == {B.BF16}, codes
    else:
        assert codes.get(B.F32_SPLIT, 0) >= 40 and codes.get(B.F32, 0) == 1, codes       # (all but the 4-channel stem)
