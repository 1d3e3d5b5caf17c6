"""A training step of SPVCNN / MinkUNet as launch plans: the whole forward pass and the whole backward pass
each cross the C-ABI ONCE (lidal_plan_run, csrc/plan.hip) instead of once per operator.

Why.  /root/reference/train.py:127-140 queues its kernels one Python call at a time, and so did rounds 1-3 of
this package (an autograd Function per block, ~230 ctypes calls, ~800 launches, a few tensor allocations per
call): on ONE ~120 k-point scan -- BASELINE.json's literal configuration -- the step took 11 ms of wall time for
~6 ms of GPU work, bound by the one Python thread.  Once the coordinate tables of a batch exist
(network/geometry.py builds them ahead of the features) every row count of the step is known, so the host can
lay the step out before its first feature kernel:

  * `_Program` (once per model): the layers of the U-Net as flat records -- which parameter, which BatchNorm
    buffers, channel counts, where each parameter's gradient lives in ONE flat f32 buffer;
  * `_Run` (once per step): walks the program and writes the operator calls of the forward pass -- the SAME
    entry points with the SAME arguments, in the SAME order as the per-operator path (nn/functional/*.py,
    network/blocks.py) -- as 64-bit words into a list, carving every activation out of a few large blocks
    (`_Arena`: bump allocation, no tensor objects, no allocator calls per activation); `lidal_plan_run`
    executes the words in one call.  The backward pass is written the same way when autograd asks for it.
  * autograd sees ONE node for the whole network (`_PlannedNet`); the cross-entropy stays the caller's
    (train.py:136).

Results are bitwise those of the per-operator path (tests/test_plan_gpu.py: loss, logits, every parameter
gradient, every BatchNorm buffer, both networks, f32 and bf16), which stays in the package as the reference
the plan is checked against (LIDAL_PLAN=0 selects it).

What still goes through torch inside a planned step: the two in-place dropouts of SPVCNN (network/spvcnn.py:
139,147: torch's generator decides the mask; the plan is cut in three around them), the flat gradient buffer and
its per-parameter views (one call), and the optimizer.
"""
import array
import ctypes
import os
import struct

import torch

from .. import backend as B
from ..nn.functional import conv as _C
from ..nn.functional import norm as _N
from ..nn.functional import devoxelize as _DV
from ..nn.functional.invlist import inverse_lists
from ..nn.functional.voxelize import _index32

__all__ = ['planned_forward', 'enabled', 'COUNTERS']

ENABLED = os.environ.get('LIDAL_PLAN', '1') != '0'
BN_SUMS = None          # None: follow network.blocks.BN_SUMS

# operation kinds of lidal_plan_run (include/lidal_amd.h; tests/test_host_cpu.py checks them against the header)
OP_CONV_WEIGHT_IMAGE_BATCH, OP_CONV_APPLY_IMAGE, OP_CONV_DGRAD_BN_SUMS, OP_CONV_WGRAD = 1, 2, 3, 4
OP_BN_TRAIN_FWD, OP_BN_TRAIN_FWD_TILES, OP_BN_BWD, OP_BN_BWD_TILES, OP_BN_EVAL_FWD, OP_BN_FOLD = 5, 6, 7, 8, 9, 10
OP_COLSUM, OP_ADD_RELU_FWD, OP_ADD_RELU_BWD, OP_VOXELIZE_FWD_1TO1, OP_VOXELIZE_FWD_SORTED = 11, 12, 13, 14, 15
OP_VOXELIZE_BWD, OP_DEVOXELIZE_FWD, OP_DEVOXELIZE_BWD_SORTED, OP_CE_FWD, OP_CE_BWD = 16, 17, 18, 19, 20
OP_COPY2D, OP_ADD2D, OP_TRANSPOSE_F32, OP_CAST_ROWS_BF16, OP_VIEW_MEAN_SOFTMAX = 21, 22, 23, 24, 25
OP_FORK_SIDE, OP_JOIN_SIDE, OP_CONV_APPLY_IMAGE_WS, OP_CONV_DGRAD_BN_SUMS_WS = 26, 27, 28, 29
OP_ADD_RELU_BWD_BN_SUMS, OP_BN_BWD_FROM_SUMS, OP_ADD_RELU_BWD_BN_TILE_SUMS, OP_DEVOXELIZE_BWD_CELLS = 30, 31, 32, 33
OP_CONV_WGRAD_STREAMS = 34

# operations executed inside plans ('ops') and plans run ('plans') since import (backend.HITS counts every
# library call made from Python, 'plan_run' among them)
COUNTERS = {'ops': 0, 'plans': 0}
# tests: also count the operations of every plan in backend.HITS under the names the per-operator path counts them
# (walks the words in Python: off in production)
TALLY = os.environ.get('LIDAL_PLAN_TALLY', '0') != '0'
# tests: a list -> every training run appends itself when its backward pass has been queued (an inference run: when its
# one plan has been queued), keeps the words of its
# plans (`tapes`: (phase, words) per lidal_plan_run call) and does NOT give its memory back, so that a test can
# replay every operation against the oracle on the operation's own stored operands (tests/test_teacher_forced_gpu.py);
# the test calls run.release() when it is done
TRACE = None
# The weight gradients of layers with at most this many rows run on a SECOND STREAM beside the data gradient and the
# BatchNorm backward that follow them in the plan (0 = off; default: all of them).  A layer's weight gradient and its
# data gradient both depend only on the gradient of its output, and on the coarse levels either kernel is latency bound
# (a launch is as long as its heaviest tile's chain of phases and leaves most of the chip idle).  Inside the plan a
# fork / join is two runtime calls in C++ (round 2's per-operator form paid stream switches in Python and lost on a
# single scan).  Measured, 3 runs each, same box: 5 scans 16.05 / 16.05 / 16.11 -> 15.42 / 15.39 / 15.38 ms,
# one scan 7.47 / 7.48 / 7.46 -> 7.01 / 7.82 / 7.02 ms (scripts/exp/side_wgrad.sh).  Same kernels, same results.
SIDE_ROWS = int(os.environ.get('LIDAL_PLAN_SIDE_ROWS', str(1 << 40)))
SIDE_MIN_ROWS = int(os.environ.get('LIDAL_PLAN_SIDE_MIN_ROWS', '0'))      # (measured: every threshold above 0 loses, matrix2.sh)
# The f32 mode keeps its weight gradients on the MAIN stream (round 6; LIDAL_PLAN_SIDE_F32=1: beside the data gradients).
# Round 5: one SPVCNN f32 run in ~20 was not bit-reproducible with the fused f64 block tail running while f32 weight gradients
# ran beside it (gradients differing in the last bits from some layer on; never under bf16).  Round 6, with the weight
# gradients in the split form (wgrad_split_kernel) on the side stream: EVERY run -- bisected (profiles/README.md, round 6) to
# the overlap of a BatchNorm backward (f64 sums) with a split-form convolution weight gradient; neither an instruction
# fault, nor the pair in isolation, nor a shared buffer, nor torch's kernels.  Two configurations are clean (0 of 48
# repetitions each): no side stream (34.8 ms per 5-scan f32 step) and F32_BN_ALONE below with the side stream (34.6); the
# unexplained overlap buys 32.2.  Until the cause is known the f32 mode -- the parity mode -- runs single-stream.
SIDE_F32 = os.environ.get('LIDAL_PLAN_SIDE_F32', '0') == '1'
# The shortcut branch of a residual block (1x1x1 convolution + BatchNorm, forward and backward) runs on a THIRD stream
# beside the block's main branch where the level has at least this many rows (0 = never): 5 scans 15.37 / 15.38 / 15.34
# -> 15.19 / 15.24 ms; on one scan the extra fork / join pairs cost the host what the overlap wins on the GPU (6.65 /
# 7.24 -> 6.81 / 7.42 ms; scripts/exp/matrix.sh, matrix2.sh), hence the threshold: the levels of a multi-scan batch.
BRANCH_ROWS = int(os.environ.get('LIDAL_PLAN_BRANCH_ROWS', '100000'))
# The backward pass of SPVCNN's point branch (BatchNorm1d backward, the Linear's two gradients, its bias gradient: streaming
# kernels over all points) beside the voxel branch, between the place its input gradient appears and the place its result
# is consumed (round 5) -- on side stream 1, the WEIGHT GRADIENTS' stream (so point_join() also waits for the weight
# gradients queued there before it; a stream of its own, index 3, was measured: the process then has more streams than
# hardware queues and the step went from 13.9 to 20.5-21.2 ms, profiles/README.md).  bf16 only (the f32 mode keeps its
# concurrency as it was, section 5 of DESIGN.md).
POINT_SIDE = int(os.environ.get('LIDAL_PLAN_POINT_SIDE', '1'))     # 0 = on the main stream, i = on side stream i
_XJOIN = set(filter(None, os.environ.get('LIDAL_X_JOIN_BEFORE', '').split(',')))
F32_BN_ALONE = os.environ.get('LIDAL_PLAN_F32_BN_ALONE', '1') != '0'
# experiment switches of the backward plan, read once (a step consulted the environment ~140 times for them: 0.1 ms of a
# host-bound single-scan step)
_X_SPLIT_CONV_OFF = os.environ.get('LIDAL_X_SPLIT_CONV') == '0'
_X_SPLIT_DENSE_OFF = os.environ.get('LIDAL_X_SPLIT_DENSE') == '0'
_X_STREAMS_MAIN = os.environ.get('LIDAL_X_STREAMS_MAIN') == '1'
_X_WGRAD_ARENA = os.environ.get('LIDAL_X_WGRAD_ARENA') == '1'
_X_JOIN_AFTER_WGRAD = os.environ.get('LIDAL_X_JOIN_AFTER_WGRAD') == '1'
_NARGS = {}
_HIT_NAMES = {1: 'conv_weight_image', 2: 'conv_apply', 3: 'conv_apply', 4: 'conv_wgrad', 5: 'bn_train_fwd',
              6: 'bn_train_fwd', 7: 'bn_bwd', 8: 'bn_bwd', 9: 'bn_eval_fwd', 10: 'bn_fold', 11: 'colsum',
              12: 'add_relu_fwd', 13: 'add_relu_bwd', 14: 'voxelize_fwd_1to1', 15: 'voxelize_fwd_sorted',
              16: 'voxelize_bwd', 17: 'devoxelize_fwd', 18: 'devoxelize_bwd_sorted', 19: 'ce_fwd', 20: 'ce_bwd',
              21: 'copy2d', 22: 'add2d', 23: 'transpose_f32', 24: 'cast_rows_bf16', 25: 'view_mean_softmax',
              26: 'fork_side', 27: 'join_side', 28: 'conv_apply', 29: 'conv_apply', 30: 'add_relu_bwd', 31: 'bn_bwd',
              32: 'add_relu_bwd', 33: 'devoxelize_bwd_cells', 34: 'conv_wgrad_streams'}


def _tally(words):
    if not _NARGS:
        L = B.lib_handle()
        for k in _HIT_NAMES:
            _NARGS[k] = int(L.lidal_plan_op_args(k))
    i, n = 0, len(words)
    while i < n:
        kind = words[i] & 0xFFFF
        name = _HIT_NAMES[kind]
        if kind in (OP_CONV_APPLY_IMAGE, OP_CONV_APPLY_IMAGE_WS) and words[i + 3] == 0:
            name = 'conv_apply(dense)'
        elif kind == OP_CONV_WGRAD and words[i + 5] == 0:
            name = 'conv_wgrad(dense)'
        elif kind == OP_BN_BWD_TILES:
            B.hit('bn_bwd(tile sums)')
        B.hit(name)
        i += 1 + _NARGS[kind]


def enabled():
    return ENABLED


def _dbits(x):
    """A float argument as the bit pattern of a double (how lidal_plan_run takes float arguments)."""
    return struct.unpack('<q', struct.pack('<d', float(x)))[0]


# ---- memory --------------------------------------------------------------------------------------------
class _Arena:
    """Bump allocation out of a few large blocks of torch's caching allocator: an activation is an address,
    not a tensor.  Blocks have ONE size -- 1 GiB: a 5-scan step's activations are three of them, 288 GB of HBM
    do not notice the slack -- so the allocator re-uses them exactly, step after step, whatever the row counts of
    the step are (per-activation tensors of ever-changing sizes fragment its pools: the reserved bytes of a
    never-repeating input stream kept growing).  A buffer of more than a quarter block gets a block of its own,
    rounded up to 64 MiB."""
    BLOCK = int(os.environ.get('LIDAL_PLAN_BLOCK_MB', '1024')) << 20

    def __init__(self, device):
        self.device = device
        self.blocks = []            # (address, size, uint8 tensor)
        self.cur = self.end = 0

    def alloc(self, nbytes):
        nbytes = (nbytes + 255) & -256
        if nbytes > (self.BLOCK >> 2):
            size = (nbytes + (64 << 20) - 1) & -(64 << 20)
            t = torch.empty(size, dtype=torch.uint8, device=self.device)
            a = t.data_ptr()
            self.blocks.append((a, size, t))
            return a
        a = self.cur
        if a + nbytes > self.end:
            t = torch.empty(self.BLOCK, dtype=torch.uint8, device=self.device)
            a = t.data_ptr()
            self.blocks.append((a, self.BLOCK, t))
            self.end = a + self.BLOCK
        self.cur = a + nbytes
        return a

    def tensor(self, addr, shape, dtype):
        """A torch view of arena memory (for what leaves the plan: logits, features, dropout operands)."""
        numel = 1
        for s in shape:
            numel *= s
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        for a, size, t in self.blocks:
            if a <= addr and addr + nbytes <= a + size:
                return t[addr - a:addr - a + nbytes].view(dtype).view(*shape)
        raise RuntimeError('lidal_amd.plan: address outside the arena')

    def release(self):
        self.blocks = []
        self.cur = self.end = 0


_SCRATCH = {}       # (device index, stream) -> [tensor, address, size]: workspace of ONE operation at a time


def _scratch(device, stream, nbytes, keep):
    """Workspace that lives for one operation (BatchNorm partials, split-K slabs, segment partials): the
    operations of a plan run one after the other on one stream, so they share one buffer per stream.  When a
    request outgrows the buffer, the old one goes to `keep` (operations already written may point into it)."""
    key = (device.index, stream)
    s = _SCRATCH.get(key)
    if s is None or s[2] < nbytes:
        if s is not None:
            keep.append(s[0])
        size = max(32 << 20, (int(nbytes * 1.25) + (1 << 20)) & -(1 << 20))
        t = torch.empty(size, dtype=torch.uint8, device=device)
        s = _SCRATCH[key] = [t, t.data_ptr(), size]
    return s[1]


# ---- the program: the model as flat records --------------------------------------------------------------
class _Conv:
    __slots__ = ('w', 'k', 'ci', 'co', 'transposed', 'strided', 'shape', 'role', 'pad_in', 'img_f', 'img_b', 'tkey',
                 'param', 'ptrs')


class _BN:
    __slots__ = ('w', 'b', 'rm', 'rv', 'nbt', 'c', 'eps', 'mom', 'relu', 'mod', 'scale', 'shift')


class _Lin:
    __slots__ = ('conv', 'b', 'ci', 'co', 'co_pad', 'bias')


class _Res:
    __slots__ = ('c1', 'b1', 'c2', 'b2', 'cs', 'bs')


class _Program:
    """Compiled once per model instance (parameters are looked up by position, so a re-assigned parameter --
    model.to(), .half() -- recompiles: `signature`)."""

    def __init__(self, model):
        from . import blocks
        from .unet import SPVCNN
        from .. import nn as spnn
        self.kind = type(model).__name__
        self.spvcnn = isinstance(model, SPVCNN)
        self.params = list(model.parameters())
        # (module, name, tensor) of every parameter and BatchNorm buffer: `valid()` re-checks them every step
        self.where = [(m, k, p) for m in model.modules() for k, p in m._parameters.items() if p is not None]
        self.where_buf = []
        self.index = {id(p): i for i, p in enumerate(self.params)}
        assert len(self.index) == len(self.params), 'shared parameters are not planned'
        self.buffers = []
        self.used = set()
        self.convs = []
        self.bns = []

        def conv(m, pad_in=0):
            c = _Conv()
            c.param = m.kernel
            c.w = self._p(m.kernel)
            c.k, c.ci, c.co = m.kernel_volume, m.in_channels, m.out_channels
            c.transposed = bool(m.transposed)
            c.strided = any(s > 1 for s in m.stride)
            c.shape = (c.k, c.ci, c.co)
            c.role = 0
            c.pad_in = pad_in
            c.img_f = c.img_b = 0
            c.tkey = None
            c.ptrs = {}
            assert m.bias is None and m.bn_follows
            self.convs.append(c)
            return c

        def bn(m, relu):
            r = _BN()
            r.w, r.b = self._p(m.weight), self._p(m.bias)
            r.rm, r.rv, r.nbt = self._b(m.running_mean), self._b(m.running_var), self._b(m.num_batches_tracked)
            self.where_buf += [(m, 'running_mean', m.running_mean), (m, 'running_var', m.running_var),
                               (m, 'num_batches_tracked', m.num_batches_tracked)]
            r.c = m.num_features
            r.eps, r.mom = _dbits(m.eps), _dbits(m.momentum)
            r.relu = int(bool(relu))
            r.mod = m
            r.scale = r.shift = 0
            assert bool(m.fused_relu) == bool(relu)
            self.bns.append((r, m))
            return r

        def conv_bn(seq, at, relu):
            return conv(seq[at]), bn(seq[at + 1], relu)

        def res(m):
            r = _Res()
            r.c1, r.b1 = conv_bn(m.net, 0, True)
            r.c2, r.b2 = conv_bn(m.net, 3, False)
            r.cs = r.bs = None
            if not isinstance(m.downsample, torch.nn.Identity):
                r.cs, r.bs = conv_bn(m.downsample, 0, False)
                assert r.cs.k == 1
            else:
                assert r.c1.ci == r.c2.co
            return r

        def lin(m, stats):
            r = _Lin()
            c = _Conv()
            c.param = m.weight
            c.w = self._p(m.weight)
            c.k, c.ci, c.co = 1, m.in_features, m.out_features
            c.transposed = c.strided = False
            c.shape = (1, c.ci, c.co)
            c.role = 1                      # nn.Linear keeps [Cout, Cin]
            c.pad_in = 0
            c.img_f = c.img_b = 0
            c.tkey = None
            c.ptrs = {}
            self.convs.append(c)
            r.conv = c
            r.b = self._p(m.bias)
            r.bias = m.bias
            r.ci, r.co = c.ci, c.co
            r.co_pad = c.co
            assert bool(m.bn_follows) == bool(stats)
            return r

        self.stem = [conv_bn(model.stem, 0, True), conv_bn(model.stem, 3, True)]
        self.stem[0][0].pad_in = 1          # 4 input channels: padded to one 16-byte vector under bf16
        self.stages = []
        for i in range(1, 5):
            st = getattr(model, 'stage%d' % i)
            self.stages.append((conv_bn(st[0].net, 0, True), res(st[1]), res(st[2])))
        self.ups = []
        for i in range(1, 5):
            up = getattr(model, 'up%d' % i)
            self.ups.append((conv_bn(up[0].net, 0, True), res(up[1][0]), res(up[1][1])))
        self.classifier = lin(model.classifier[0], False)
        self.n_class = self.classifier.co
        self.points = []
        self.dropout = None
        if self.spvcnn:
            for seq in model.point_transforms:
                self.points.append((lin(seq[0], True), bn(seq[1], True)))
            self.dropout = model.dropout
        assert self.used == set(range(len(self.params))), 'a parameter of the model is not part of the plan'
        # where each parameter's gradient lives in the flat f32 buffer (whole 16-byte vectors; the odd-sized ones last,
        # and the classifier's bias ALWAYS the very last: its column sum is taken over the padded logits' columns and
        # writes up to 7 floats past its slot -- into the buffer's tail padding, never into another gradient, whatever
        # the class count is (19: odd-sized anyway; 12 / 20 / 28 under bf16: multiples of 4 but not of 8))
        cls_b = self.classifier.b
        order = sorted(range(len(self.params)), key=lambda i: (i == cls_b, self.params[i].numel() % 4 != 0, i))
        assert order[-1] == cls_b
        self.slot = [0] * len(self.params)
        off = 0
        self.flat_order = order
        for i in order:
            self.slot[i] = off
            off += self.params[i].numel()
            if self.params[i].numel() % 4 and i != order[-1]:
                off = (off + 3) & -4
        self.flat_numel = off
        self.flat_exact = all(self.params[i].numel() % 4 == 0 for i in order[:-1])
        self.flat_params = None
        self.running = [t for r, m in self.bns for t in (m.running_mean, m.running_var)]
        self.tensors = self.params + self.buffers
        self.bn_modules = [m for _, m in self.bns]
        assert len(self.where) == len(self.params)
        self.ones = None
        self.cls_shift = None

    def _p(self, p):
        i = self.index[id(p)]
        assert i not in self.used, 'a parameter is used twice'
        self.used.add(i)
        return i

    def _b(self, t):
        self.buffers.append(t)
        return len(self.params) + len(self.buffers) - 1

    def valid(self, train=True):
        """Is the model still the one this program was compiled from, in the configuration the plan covers?
        (Parameters can be re-assigned, cast, moved or frozen, buffers are REPLACED by Module._apply, modules can be
        put in eval / train mode one by one: checked every step, ~50 us.)"""
        f32 = torch.float32
        dev = self.params[0].device
        if not (dev.type == 'cuda'
                and all(m._parameters.get(k) is p and p.dtype is f32 and p.device == dev for m, k, p in self.where)
                and all(m._buffers.get(k) is t for m, k, t in self.where_buf)):
            return False
        if train:
            return (all(p.requires_grad for p in self.params)
                    and all(m.training and m.track_running_stats for m in self.bn_modules))
        return not any(m.training for m in self.bn_modules) and all(m.track_running_stats for m in self.bn_modules)

    def eval_operands(self, dev):
        """Inference: the folded BatchNorm maps (blocks._fold: cached on the modules) and the point-branch shifts
        `shift + bias * scale` (dense.py _forward) of the current parameters -- addresses, recomputed only when a
        parameter or buffer has changed (version counters, weight epoch)."""
        from .blocks import _fold
        stamp = (B.WEIGHT_EPOCH[0], str(dev)) + tuple(t._version for t in self.tensors)
        if getattr(self, '_eval_stamp', None) == stamp:
            return
        self._eval_keep = []
        for r, m in self.bns:
            sc, sh = _fold(m, dev)
            r.scale, r.shift = sc.data_ptr(), sh.data_ptr()
            self._eval_keep += [sc, sh]
        self._point_shift = []
        for lin, r in self.points:
            sc, sh = _fold(r.mod, dev)
            t = sh + lin.bias.detach().float() * sc
            self._eval_keep.append(t)
            self._point_shift.append(t.data_ptr())
        self._eval_stamp = stamp


def _program(model, train=True):
    """The compiled program of `model`, or None if the model is not in the planned configuration."""
    prog = model.__dict__.get('_lidal_program')
    if prog is not None and prog.valid(train):
        return prog
    try:
        ok = all(p.dtype == torch.float32 and p.is_cuda and p.is_contiguous() for p in model.parameters())
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                ok = ok and m.track_running_stats and m.momentum is not None and m.affine
        prog = _Program(model) if ok else None
        if prog is not None and not prog.valid(train):
            return None                 # (compiled, but not in this mode's configuration right now)
    except (AssertionError, AttributeError, KeyError):
        prog = None
    model.__dict__['_lidal_program'] = prog
    return prog


def plannable(model, x):
    """'train' / 'eval' if this forward pass is one of the two standard configurations the plans cover (a training
    step with gradients; inference under no_grad in eval mode), else None.  (Anything else runs the per-operator
    path: same results, one Python call per operator.)"""
    from . import blocks
    if not ENABLED:
        return None
    if model.training and torch.is_grad_enabled():
        if not (blocks.FUSE_BLOCKS and B.FORK == 15):
            return None
        mode = 'train'
    elif not model.training and not torch.is_grad_enabled():
        mode = 'eval'
    else:
        return None
    f = x.F
    if not (torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32 and f.dim() == 2 and f.shape[1] == 4
            and f.is_contiguous() and not f.requires_grad and f.shape[0] > 1 and tuple(x.s) == (1, 1, 1)):
        return None
    if torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') != torch.bfloat16:
        return None
    return mode if _program(model, mode == 'train') is not None else None


# ---- one step ---------------------------------------------------------------------------------------------
_WS_BN = {}
_WS_SLABS = {}
_WS_SEG = {}


def _bn_ws(n, c):
    key = (n, c)
    v = _WS_BN.get(key)
    if v is None:
        if len(_WS_BN) > 4096:
            _WS_BN.clear()
        v = _WS_BN[key] = int(B.lib_handle().lidal_bn_workspace_bytes(n, c))
    return v


def _slabs(n_a, n_b, k, ca, cb, dtype):
    """(dtype code, slabs) of a weight gradient, memoised: conv.wgrad_plan (the rule of the per-operator path)."""
    key = (n_a, n_b, k, ca, cb, dtype, B.SPLIT_F32_TRAIN, B.SPLIT_F32)
    v = _WS_SLABS.get(key)
    if v is None:
        if len(_WS_SLABS) > 4096:
            _WS_SLABS.clear()
        v = _WS_SLABS[key] = _C.wgrad_plan(n_a, n_b, k, ca, cb, dtype)
    return v


_WS_STREAM = {}


def _stream_serves(n, k, ca, cb):
    """lidal_conv_wgrad_streams_serves, memoised."""
    key = (n, k, ca, cb)
    v = _WS_STREAM.get(key)
    if v is None:
        if len(_WS_STREAM) > 4096:
            _WS_STREAM.clear()
        v = _WS_STREAM[key] = bool(B.lib_handle().lidal_conv_wgrad_streams_serves(n, n, k, ca, cb))
    return v


def _seg_ws(n_entries, m, c):
    key = (n_entries, m, c)
    v = _WS_SEG.get(key)
    if v is None:
        if len(_WS_SEG) > 4096:
            _WS_SEG.clear()
        v = _WS_SEG[key] = int(B.lib_handle().lidal_segment_workspace_bytes(n_entries, m, c))
    return v


class _Tables:
    """The addresses a step reads out of its Geometry, gathered once per geometry."""

    def __init__(self, g, spvcnn, train, model_cs=None):
        x0 = g.x0
        km, cm = x0.kmaps, x0.cmaps
        self.n = []
        self.k3 = []        # per level: (table, perm, masks, nbmaps, koff, spairs, sdesc, n_wg) -- the last three 0 without stream tables
        self.k2 = []        # per level l -> l + 1: (out table, perm, masks, in table, perm, masks, nbmaps, koff)
        self.keep = []
        for l in range(5):
            s = 1 << l
            st = (s, s, s)
            self.n.append(int(cm[st].shape[0]))
            k3 = km[(st, (3, 3, 3), (1, 1, 1), (1, 1, 1))]
            oo = k3.order_out
            st3 = k3._streams if (train and k3._streams is not None) else None       # (built with the geometry: conv.prefetch_kernel_maps)
            self.k3.append((oo.table.data_ptr(), oo.perm.data_ptr(), oo.tile_masks.data_ptr(),
                            k3._nbmaps_cap.data_ptr() if train else 0, k3.koff.data_ptr() if train else 0,
                            st3[0].data_ptr() if st3 else 0, st3[1].data_ptr() if st3 else 0, st3[2] if st3 else 0))
            self.keep.append(k3)
            if l < 4:
                k2 = km[(st, (2, 2, 2), (2, 2, 2), (1, 1, 1))]
                oo, oi = k2.order_out, k2.order_in
                self.k2.append((oo.table.data_ptr(), oo.perm.data_ptr(), oo.tile_masks.data_ptr(),
                                oi.table.data_ptr(), oi.perm.data_ptr(), oi.tile_masks.data_ptr(),
                                k2._nbmaps_cap.data_ptr() if train else 0, k2.koff.data_ptr() if train else 0))
                self.keep.append(k2)
        self.pt = {}
        self.p = 0
        if spvcnn:
            z = g.z
            af = z.additional_features
            self.p = int(z.C.shape[0])
            for l, key in ((0, 1), (4, (16, 16, 16)), (2, (4, 4, 4))):
                if l == 0 and af['idx_query'].get((1, 1, 1)) is not None:
                    key_v = (1, 1, 1)
                else:
                    key_v = key
                idx64, counts = af['idx_query'][key_v], af['counts'][key_v]
                m = self.n[l]
                idx32 = _index32(idx64)
                one = bool(getattr(idx64, '_lidal_one_to_one', False)) and idx64.numel() == m
                vorder = vseg = 0
                if not one:
                    o, sp = inverse_lists(idx32, m)
                    vorder, vseg = o.data_ptr(), sp.data_ptr()
                    self.keep += [o, sp]
                s = 1 << l
                idx8, w8 = z.idx_query[(s, s, s)], z.weights[(s, s, s)]
                assert idx8.dtype == torch.int32 and idx8.is_contiguous() and w8.dtype == torch.float32
                # (backward only) the per-voxel contributor lists -- or, on a level with many points per voxel, the
                # per-cell lists of F.devoxelize.devox_cells: the rule of the per-operator path, cells_mode
                pc = {0: model_cs[8], 4: model_cs[4], 2: model_cs[6]}[l]
                cells = None
                do = dsp = idx8
                if train and _DV.cells_mode(idx8, self.p, m, pc):
                    if getattr(idx8, '_lidal_cell_index', None) is None:
                        idx8._lidal_cell_index = idx32
                    cells = _DV.devox_cells(idx8, m)
                elif train:
                    do, dsp = inverse_lists(idx8, m, w8)
                cnt = counts if counts.dtype == torch.int32 and counts.is_contiguous() else counts.contiguous().int()
                self.keep += [idx32, cnt, idx8, w8, do, dsp, cells]
                self.pt[l] = (idx32.data_ptr(), cnt.data_ptr(), int(one), vorder, vseg,
                              idx8.data_ptr(), w8.data_ptr(), do.data_ptr(), dsp.data_ptr(),
                              tuple(t.data_ptr() for t in cells) if cells is not None else None)


def _tables(g, spvcnn, train, model_cs=None):
    key = '_plan_tables_train' if train else '_plan_tables'
    t = g.__dict__.get(key) or (g.__dict__.get('_plan_tables_train') if not train else None)
    if t is None:
        t = g.__dict__[key] = _Tables(g, spvcnn, train, model_cs)
    return t


class _Run:
    """One training step: the forward plan, what it saved, the backward plan."""
    TRAIN = True

    def __init__(self, model, prog, geometry, feats, code):
        self.prog = prog
        self.model = model
        self.dev = feats.device
        self.code = code
        self.esz = 2 if code == B.BF16 else 4
        self.vec = 8 if code == B.BF16 else 4
        self.bf16 = code == B.BF16
        self.dtype = torch.bfloat16 if self.bf16 else torch.float32
        self.geometry = geometry
        self.T = _tables(geometry, prog.spvcnn, self.TRAIN, getattr(model, 'cs', None))
        self.feats = feats
        self.stream = B.stream()
        self.arena = _Arena(self.dev)
        self.barena = None
        self.tile = B.stats_tile_rows()
        from . import blocks
        self.bn_sums = (blocks.BN_SUMS if BN_SUMS is None else BN_SUMS) and self.bf16
        self.saved = {}
        self.noise = []
        self.keep = []
        self.tapes = [] if TRACE is not None else None
        self.sides = {}                     # side stream index -> raw stream
        self.open = set()                   # side streams forked and not joined yet
        self.w = []
        self.nops = 0
        self.done = False
        self.ptr = [t.data_ptr() for t in prog.tensors]
        self._constants()
        self._images()

    # -- small persistent operands ---------------------------------------------------------------------
    def _constants(self):
        prog, T = self.prog, self.T
        if prog.ones is None or prog.ones.device != self.dev:
            prog.ones = torch.ones(512, dtype=torch.float32, device=self.dev)
            prog.cls_shift = torch.zeros(64, dtype=torch.float32, device=self.dev)
        self.ones = prog.ones.data_ptr()
        self.cls_shift = prog.cls_shift.data_ptr()
        # rule offsets [0, n] of the identity rule list, for every row count a dense weight gradient sees: one upload
        ns = [T.p] + T.n if prog.spvcnn else T.n
        host = []
        for n in ns:
            host += [0, n]
        k = torch.tensor(host, dtype=torch.int64).to(self.dev, non_blocking=True)
        self.koff_t = k
        base = k.data_ptr()
        self.koff = {n: base + 16 * i for i, n in reversed(list(enumerate(ns)))}

    def ccode(self, c):
        """dtype code of the forward product and the data gradient of layer `c` (a training run): the features' own --
        or, in f32, the split form for the sparse convolutions whose two channel counts are whole 32-channel slices
        (backend.conv_code: the rule of the per-operator path)."""
        if c.k > 1 and not c.role:
            return B.conv_code(self.dtype, c.ci, False, c.co)
        return self.code

    def _images(self):
        """The LDS images of every weight: registered with the step's bank (nn/functional/conv.py _ImageBank) under
        the tilings this step's row counts select, rebuilt by ONE launch."""
        prog, T = self.prog, self.T
        bank = _C._IMAGE_BANK
        code, dtype = self.code, self.dtype
        seen = []
        for c, (nf, nb) in self._conv_rows():
            code = self.ccode(c)
            key = (_C._tiling(c.ci, c.co, code, nf), _C._tiling(c.co, c.ci, code, nb), code)
            e = c.tkey.get(key) if c.tkey else None
            if e is None or e['ref']() is not c.param or bank.entries.get((id(c.param), e['key'])) is not e:
                bank.get(c.param, dtype, nf, nb, c.shape if (c.role or c.k == 1) else None, c.role, code)
                e = bank.entry(c.param, key)
                if c.tkey is None or len(c.tkey) > 8:
                    c.tkey = {}
                c.tkey[key] = e
                c.ptrs = {}
            p = c.ptrs.get(key)
            if p is None:
                p = c.ptrs[key] = (e['img_f'].data_ptr(), e['img_b'].data_ptr())
            c.img_f, c.img_b = p
            e['used'] = bank.tick
            seen.append((e, c.param))
        # one launch rebuilds every stale image of the group (bank.get does it when it meets a stale entry; when no
        # layer needed registering, ask for it here).  Staleness is decided weight by weight, as bank.get does on the
        # per-operator path: a load_state_dict(strict=False) or a copy_ into ONE layer between two forward passes moves
        # that layer's version counter only
        # (an f32 inference run has two groups: the split-form images and the exact-f32 image of the 4-channel stem)
        for e, w in seen:
            if e['version'] != B.weights_key(w):
                bank._rebuild(e['group'])

    def _conv_rows(self):
        """(layer, (rows its forward produces, rows its data gradient produces)) for every weight of the program."""
        prog, n = self.prog, self.T.n
        out = []
        for cb in prog.stem:
            out.append((cb[0], (n[0], n[0])))
        for l, (down, ra, rb) in enumerate(prog.stages):
            out.append((down[0], (n[l + 1], n[l])))
            for r in (ra, rb):
                for c in (r.c1, r.c2, r.cs):
                    if c is not None:
                        out.append((c, (n[l + 1], n[l + 1])))
        for i, (dec, ra, rb) in enumerate(prog.ups):
            l = 3 - i                               # output level of up_{i+1}
            out.append((dec[0], (n[l], n[l + 1])))
            for r in (ra, rb):
                for c in (r.c1, r.c2, r.cs):
                    if c is not None:
                        out.append((c, (n[l], n[l])))
        rows = self.T.p if prog.spvcnn else n[0]
        out.append((prog.classifier.conv, (rows, rows)))
        for lin, _ in prog.points:
            out.append((lin.conv, (self.T.p, self.T.p)))
        return out

    # -- running a stretch of the tape --------------------------------------------------------------------
    def flush(self):
        if not self.nops:
            return
        for which in sorted(self.open):     # every stretch of the plan ends with the main stream waiting for the side streams
            self.join(which)
        if TALLY:
            _tally(self.w)
        if self.tapes is not None:
            self.tapes.append(('bwd' if self.barena is not None else 'fwd', list(self.w)))
        arr = array.array('q', self.w)
        addr, n_words = arr.buffer_info()
        L = B.lib_handle()
        if self.sides:
            n = max(self.sides) + 1
            streams = (ctypes.c_void_p * n)(*([self.stream] + [self.sides.get(i) for i in range(1, n)]))
            rc = L.lidal_plan_run_streams(addr, n_words, self.nops, streams, n)
        else:
            rc = L.lidal_plan_run(addr, n_words, self.nops, self.stream, None)
        COUNTERS['plans'] += 1
        COUNTERS['ops'] += self.nops
        B.HITS['plan_run'] = B.HITS.get('plan_run', 0) + 1
        self.w = []
        self.nops = 0
        if rc != 0:
            raise RuntimeError('lidal_amd.plan_run failed (%d): %s' % (rc, L.lidal_last_error().decode()))

    def scratch(self, nbytes, flag=0):
        """Workspace of one operation on the main stream, or (flag = a side stream's) on that side stream."""
        return _scratch(self.dev, (self.stream, flag) if flag else self.stream, nbytes, self.keep)

    def fork(self, which):
        """Side stream `which` waits for what the plan has queued on the main stream so far; -> the flag of its operations."""
        if which not in self.sides:
            self.sides[which] = B.side_stream(self.dev, which).cuda_stream
        self.w.append(OP_FORK_SIDE | (which << 16))
        self.nops += 1
        self.open.add(which)
        return which << 16

    def join(self, which):
        self.w.append(OP_JOIN_SIDE | (which << 16))
        self.nops += 1
        self.open.discard(which)

    def _xjoin(self, what):
        """The main stream waits for the side streams before an operation of kind `what`.
        f32 mode, 'bn' (round 6, F32_BN_ALONE): a BatchNorm backward never runs while a weight gradient in the split form
        is resident beside it -- the one overlap in which the planned f32 step was seen to lose its run-to-run bit-equality
        (profiles/README.md, round 6: 100 % of the runs with the overlap, 0 of 48 without; neither kernel differs from its
        solo result when the pair is run in isolation).  LIDAL_X_JOIN_BEFORE=bn,dgrad,tail: the experiment's switches."""
        if self.open and (what in _XJOIN or (what == 'bn' and F32_BN_ALONE and not self.bf16)):
            for w_ in sorted(self.open):
                self.join(w_)

    def side(self, rows):
        """Flag of a weight gradient over `rows` rows: side stream 1 (after a fork), or 0 = the main stream."""
        if not SIDE_ROWS or rows > SIDE_ROWS or rows < SIDE_MIN_ROWS:
            return 0
        if not self.bf16 and (not SIDE_F32 or _N.TAIL_SUMS_ROWS > 0):
            # (the f32 mode: see SIDE_F32.  The one pair that was ever seen to break run-to-run bit-equality -- the fused
            # f64 block tail while f32 weight gradients run beside it -- cannot be put together by the environment knobs:
            # asking for the tail takes the weight gradients back to the main stream)
            return 0
        return self.fork(1)

    # ===================================== forward ========================================================
    def f_conv(self, c, x, n_in, ci, table, n_out, stats):
        """conv.py _apply: -> (out, tile statistics or 0)."""
        A = self.arena
        out = A.alloc(n_out * c.co * self.esz)
        st = A.alloc(-(-n_out // self.tile) * c.co * 12) if (stats and self.bf16) else 0
        wb = _C.apply_workspace_bytes(n_out, c.co)
        self.w += (OP_CONV_APPLY_IMAGE_WS, x, c.img_f, table[0], table[1], table[2], out, n_in, n_out, ci, c.co, c.k, 0,
                   self.ccode(c), 0, 0, 0, 0, st, self.scratch(wb) if wb else 0, wb)
        self.nops += 1
        return out, st

    def f_dense(self, c, x, n, co, shift, stats, flag=0):
        """dense.py _rows_gemm (role 0): x [n, ci] @ W (+ shift) -> (out [n, co], tile statistics or 0)."""
        A = self.arena
        out = A.alloc(n * co * self.esz)
        st = A.alloc(-(-n // self.tile) * co * 12) if (stats and self.bf16) else 0
        self.w += (OP_CONV_APPLY_IMAGE | flag, x, c.img_f, 0, 0, 0, out, n, n, c.ci, co, 1, 0, self.ccode(c),
                   self.ones if shift else 0, shift, 0, 0, st)
        self.nops += 1
        return out, st

    def f_bn(self, r, x, n, st, residual=0, relu_after=False, flag=0):
        """norm.py train_forward: -> (y, mean, invstd)."""
        A, p = self.arena, self.ptr
        c = r.c
        y = A.alloc(n * c * self.esz)
        mean = A.alloc(c * 4)
        invstd = A.alloc(c * 4)
        relu = r.relu | (2 if (relu_after and residual) else 0)
        if st:
            self.w += (OP_BN_TRAIN_FWD_TILES | flag, x, self.code, n, c, p[r.w], p[r.b], r.eps, r.mom, p[r.rm], p[r.rv],
                       p[r.nbt], relu, residual, y, mean, invstd, st, -(-n // self.tile))
        else:
            nb = _bn_ws(n, c)
            self.w += (OP_BN_TRAIN_FWD | flag, x, self.code, n, c, p[r.w], p[r.b], r.eps, r.mom, p[r.rm], p[r.rv],
                       p[r.nbt], relu, residual, y, mean, invstd, self.scratch(nb, flag), nb)
        self.nops += 1
        return y, mean, invstd

    def f_conv_bn(self, cb, x, n_in, table, n_out, ci=None):
        """blocks._ConvNormAct.forward; saves (x, conv output, mean, invstd)."""
        c, r = cb
        x1, st = self.f_conv(c, x, n_in, c.ci if ci is None else ci, table, n_out, True)
        y, mean, invstd = self.f_bn(r, x1, n_out, st)
        self.saved[id(cb)] = (x, x1, mean, invstd)
        return y

    def f_res(self, r, x, n, table):
        """blocks._Residual.forward."""
        fl = 0
        if r.cs is not None:                # the shortcut convolution + BatchNorm: independent of the main branch
            fl = self.fork(2) if (BRANCH_ROWS and n >= BRANCH_ROWS) else 0
            xs, sts = self.f_dense(r.cs, x, n, r.cs.co, 0, True, fl)
            res, means, invs = self.f_bn(r.bs, xs, n, sts, flag=fl)
        else:
            xs = means = invs = 0
            res = x
        x1, st1 = self.f_conv(r.c1, x, n, r.c1.ci, table, n, True)
        y1, mean1, inv1 = self.f_bn(r.b1, x1, n, st1)
        x2, st2 = self.f_conv(r.c2, y1, n, r.c2.ci, table, n, True)
        if fl:
            self.join(2)
        out, mean2, inv2 = self.f_bn(r.b2, x2, n, st2, res, True)
        self.saved[id(r)] = (x, x1, mean1, inv1, y1, x2, mean2, inv2, out, xs, means, invs)
        return out

    def f_cat(self, a, ca, b, cb, n):
        A, e = self.arena, self.esz
        out = A.alloc(n * (ca + cb) * e)
        pitch = (ca + cb) * e
        self.w += (OP_COPY2D, a, ca * e, out, pitch, n, ca * e, 0,
                   OP_COPY2D, b, cb * e, out + ca * e, pitch, n, cb * e, 0)
        self.nops += 2
        return out

    def f_devox(self, x, lvl, c):
        """devoxelize.py DevoxelizeFunction.forward at level `lvl`: [n_lvl, c] -> [P, c]."""
        t = self.T.pt[lvl]
        P = self.T.p
        out = self.arena.alloc(P * c * self.esz)
        self.w += (OP_DEVOXELIZE_FWD, x, t[5], t[6], out, P, self.T.n[lvl], c, self.code)
        self.nops += 1
        return out

    def f_vox(self, z, lvl, c, code=None):
        """voxelize.py VoxelizeFunction.forward at level `lvl`: [P, c] -> [n_lvl, c]."""
        code = self.code if code is None else code
        esz = 2 if code == B.BF16 else 4
        t = self.T.pt[lvl]
        P, m = self.T.p, self.T.n[lvl]
        out = self.arena.alloc(m * c * esz)
        if t[2]:
            self.w += (OP_VOXELIZE_FWD_1TO1, z, t[0], out, P, c, code)
        else:
            nb = _seg_ws(P, m, c)
            self.w += (OP_VOXELIZE_FWD_SORTED, z, t[3], t[4], t[1], out, m, c, code, P, self.scratch(nb) if nb else 0, nb)
        self.nops += 1
        return out

    def f_point_lin(self, pt, z_in):
        """The Linear of f_point, queued as soon as its input exists -- on the point branch's side stream (POINT_SIDE, bf16
        training): it needs nothing of the voxel branch, the BatchNorm that follows it does (the residual)."""
        lin, r = pt
        fl = self.point_fork()
        x1, st = self.f_dense(lin.conv, z_in, self.T.p, lin.co, self.ptr[lin.b], True, fl)
        return x1, st

    def f_point(self, pt, z_in, devox_out, pre=None):
        """Linear -> BatchNorm1d(+ReLU) + residual (network/spvcnn.py:136,143,151; blocks.ConvNormSequential)."""
        lin, r = pt
        P = self.T.p
        x1, st = pre if pre is not None else self.f_dense(lin.conv, z_in, P, lin.co, self.ptr[lin.b], True)
        self.point_join()
        y, mean, invstd = self.f_bn(r, x1, P, st, devox_out, False)
        self.saved[id(pt)] = (z_in, x1, mean, invstd)
        return y

    def f_classifier(self, x, n):
        lin = self.prog.classifier
        co_pad = lin.co + (-lin.co) % self.vec         # (kept with the RUN: a bf16 and an f32 pass may both be live)
        # shift = the bias padded with zeros (dense.py _forward); the padding columns of the persistent buffer stay zero
        self.w += (OP_COPY2D, self.ptr[lin.b], lin.co * 4, self.cls_shift, co_pad * 4, 1, lin.co * 4, 0)
        self.nops += 1
        out, _ = self.f_dense(lin.conv, x, n, co_pad, self.cls_shift, False)
        self.saved['cls'] = (x, n, co_pad)
        return out, co_pad

    def f_dropout(self, addr, n, c):
        """nn.Dropout(p, inplace=True) on arena memory, with torch's own kernels and generator (ATen Dropout.cpp
        _dropout_impl, in-place form): noise = empty_like(x).bernoulli_(1 - p).div_(1 - p); x.mul_(noise)."""
        d = self.prog.dropout
        if d is None or not d.training or d.p == 0:
            self.noise.append(None)
            return
        self.flush()
        x = self.arena.tensor(addr, (n, c), self.dtype)
        if d.p == 1:
            x.zero_()
            self.noise.append(0)
            return
        noise = torch.empty_like(x).bernoulli_(1 - d.p).div_(1 - d.p)
        x.mul_(noise)
        self.noise.append(noise)

    def forward(self):
        prog, T, A = self.prog, self.T, self.arena
        n = T.n
        k3, k2 = T.k3, T.k2
        bf = self.bf16
        x = self.feats.data_ptr()
        if prog.spvcnn:                     # geometry.enter: the input voxelised on its own coordinates (f32 rows)
            x = self.f_vox(x, 0, 4, B.F32)
        c0 = prog.stem[0][0]
        if bf:                              # conv.py _forward: cast, channels padded to one 16-byte vector
            xc = A.alloc(n[0] * 8 * 2)
            self.w += (OP_CAST_ROWS_BF16, x, 4, xc, 8, n[0])
            self.nops += 1
            x, ci0 = xc, 8
        else:
            ci0 = 4
        y = self.f_conv_bn(prog.stem[0], x, n[0], k3[0], n[0], ci0)
        x0 = self.f_conv_bn(prog.stem[1], y, n[0], k3[0], n[0])
        skips = [x0]
        cs = [c0.co]
        if prog.spvcnn:
            z0 = self.f_devox(x0, 0, cs[0])
            pre = self.f_point_lin(prog.points[0], z0) if self.TRAIN else None
            y = self.f_vox(z0, 0, cs[0])
        else:
            y = x0
        for l, (down, ra, rb) in enumerate(prog.stages):
            y = self.f_conv_bn(down, y, n[l], k2[l][0:3], n[l + 1])
            y = self.f_res(ra, y, n[l + 1], k3[l + 1])
            y = self.f_res(rb, y, n[l + 1], k3[l + 1])
            skips.append(y)
            cs.append(rb.c2.co)
        z = None
        if prog.spvcnn:
            z1 = self.f_point(prog.points[0], z0, self.f_devox(y, 4, cs[4]), pre)
            pre = self.f_point_lin(prog.points[1], z1) if self.TRAIN else None
            y = self.f_vox(z1, 4, cs[4])
            self.f_dropout(y, n[4], cs[4])
            z = z1
        for i, (dec, ra, rb) in enumerate(prog.ups):
            l = 3 - i
            y = self.f_conv_bn(dec, y, n[l + 1], k2[l][3:6], n[l])
            y = self.f_cat(y, dec[0].co, skips[l], cs[l], n[l])
            y = self.f_res(ra, y, n[l], k3[l])
            y = self.f_res(rb, y, n[l], k3[l])
            c_out = rb.c2.co
            if prog.spvcnn and i == 1:
                z2 = self.f_point(prog.points[1], z, self.f_devox(y, 2, c_out), pre)
                pre = self.f_point_lin(prog.points[2], z2) if self.TRAIN else None
                y = self.f_vox(z2, 2, c_out)
                self.f_dropout(y, n[2], c_out)
                z = z2
        if prog.spvcnn:
            feat = self.f_point(prog.points[2], z, self.f_devox(y, 0, c_out), pre)
            rows = T.p
        else:
            feat, rows = y, n[0]
        logits, co_pad = self.f_classifier(feat, rows)
        self.flush()
        if self.TRAIN:                      # (the kernels wrote the running statistics through raw pointers)
            torch.autograd.graph.increment_version(prog.running)
        self.c_feat = c_out
        logits_t = A.tensor(logits, (rows, co_pad), self.dtype)[:, :prog.n_class]
        feat_t = A.tensor(feat, (rows, c_out), self.dtype)
        return logits_t, feat_t

    # ===================================== backward =======================================================
    def galloc(self, nbytes):
        return self.barena.alloc(nbytes)

    def slot(self, i):
        return self.flat + 4 * self.prog.slot[i]

    def b_bn(self, r, x, n, mean, invstd, g, g_stride, sums=0, relu=None, flag=0, part=None, nparts=0):
        """norm.py train_backward (without the mask_from part): -> dx.  part: (address, bytes) of the partial sums
        lidal_add_relu_bwd_bn_sums left for this layer."""
        p = self.ptr
        c = r.c
        if not flag:
            self._xjoin('bn')
        dx = self.galloc(n * c * self.esz)
        relu = r.relu if relu is None else relu
        if part is not None and part[0]:
            self.w += (OP_BN_BWD_FROM_SUMS | flag, x, g, g_stride, self.code, n, c, p[r.w], p[r.b], relu, mean, invstd, dx,
                       self.slot(r.w), self.slot(r.b), part[0], part[1])
        elif sums:
            self.w += (OP_BN_BWD_TILES | flag, x, g, g_stride, self.code, n, c, p[r.w], p[r.b], relu, mean, invstd, dx,
                       self.slot(r.w), self.slot(r.b), sums, nparts or -(-n // self.tile))
        else:
            nb = _bn_ws(n, c)
            self.w += (OP_BN_BWD | flag, x, g, g_stride, self.code, n, c, p[r.w], p[r.b], relu, mean, invstd, dx,
                       self.slot(r.w), self.slot(r.b), self.scratch(nb, flag), nb)
        self.nops += 1
        return dx

    def b_wgrad(self, c, x, n_x, g, n_g, rules, ci=None):
        """conv.py conv_backward's wgrad(): gw [k, ci, co] f32 straight into the parameter's gradient slot."""
        ci = c.ci if ci is None else ci
        wcode, slabs = _slabs(n_x, n_g, c.k, ci, c.co, self.dtype)
        if _X_SPLIT_CONV_OFF and wcode == B.F32_SPLIT:
            wcode = self.code
            slabs = int(B.lib_handle().lidal_conv_wgrad_slabs(n_x, n_g, c.k, ci, c.co, wcode))
        flag = self.side(max(n_x, n_g))
        # the streamed form where the level has stream tables (conv.KernelMap.streams_serve: bf16, one channel tile)
        streams = (len(rules) >= 5 and rules[2] and wcode == B.BF16 and not c.transposed and n_x == n_g
                   and _stream_serves(n_x, c.k, ci, c.co))
        if streams:
            slabs = 2 * rules[4]
            if _X_STREAMS_MAIN:       # (experiment: the streamed launches on the main stream)
                flag = 0
        nbytes = slabs * ci * c.co * 4 + (c.k * ci * c.co * 4 if ci != c.ci else 0)
        partial = self.galloc(nbytes) if _X_WGRAD_ARENA else self.scratch(nbytes, flag)
        gw = self.slot(c.w)
        if ci != c.ci:                      # the channel-padded stem: gw[:, :ci_w] of the padded gradient
            gw = partial + slabs * ci * c.co * 4
        if streams:
            self.w += (OP_CONV_WGRAD_STREAMS | flag, x, g, n_x, n_g, rules[2], rules[3], rules[4], 0, gw, partial, slabs,
                       c.k, ci, c.co, wcode)
        else:
            self.w += (OP_CONV_WGRAD | flag, x, g, n_x, n_g, rules[0], rules[1], 1 if c.transposed else 0, gw, partial, slabs,
                       c.k, ci, c.co, wcode)
        self.nops += 1
        if ci != c.ci:
            self.w += (OP_COPY2D | flag, gw, ci * c.co * 4, self.slot(c.w), c.ci * c.co * 4, c.k, c.ci * c.co * 4, 0)
            self.nops += 1
        if _X_JOIN_AFTER_WGRAD and flag:
            self.join(flag >> 16)

    def b_dgrad(self, c, g, n_g, table, n_out, kflip, skip=0, bnb=None):
        """conv.py conv_backward's data gradient: -> (gin [n_out, ci], tile sums or 0)."""
        self._xjoin('dgrad')
        gin = self.galloc(n_out * c.ci * self.esz)
        wb = _C.apply_workspace_bytes(n_out, c.ci)
        ws = self.scratch(wb) if wb else 0
        if bnb is not None:
            sums = self.galloc(-(-n_out // self.tile) * c.ci * 8)
            self.w += (OP_CONV_DGRAD_BN_SUMS_WS, g, c.img_b, table[0], table[1], table[2], gin, n_g, n_out, c.co, c.ci,
                       c.k, kflip, self.code) + bnb + (sums, ws, wb)
            self.nops += 1
            return gin, sums
        self.w += (OP_CONV_APPLY_IMAGE_WS, g, c.img_b, table[0], table[1], table[2], gin, n_g, n_out, c.co, c.ci, c.k,
                   kflip, self.ccode(c), 0, 0, 0, skip, 0, ws, wb)
        self.nops += 1
        return gin, 0

    def b_dense(self, c, x, n, g, cg, need_gx=True, linear_slot=None, branch=0):
        """dense.py rows_backward (weight gradient x^T g, then the data gradient g W^T): -> gx or 0.
        `branch`: the flag of the side stream the whole layer runs on (0: data gradient on the main stream, weight
        gradient per side())."""
        ca, cb = c.ci, cg
        wcode, slabs = _slabs(n, n, 1, ca, cb, self.dtype)
        if _X_SPLIT_DENSE_OFF and wcode == B.F32_SPLIT:
            wcode = self.code
            slabs = int(B.lib_handle().lidal_conv_wgrad_slabs(n, n, 1, ca, cb, wcode))
        direct = linear_slot is None and cb == c.co
        flag = branch or self.side(n)
        nbytes = slabs * ca * cb * 4 + (0 if direct else ca * cb * 4)
        sc = self.galloc(nbytes) if _X_WGRAD_ARENA else self.scratch(nbytes, flag)
        gw = self.slot(c.w) if direct else sc + slabs * ca * cb * 4
        self.w += (OP_CONV_WGRAD | flag, x, g, n, n, 0, self.koff[n], 0, gw, sc, slabs, 1, ca, cb, wcode)
        self.nops += 1
        if not direct:                      # nn.Linear: [Cout, Cin] = (x^T g)[:, :Cout]^T
            self.w += (OP_TRANSPOSE_F32 | flag, gw, cb, self.slot(c.w), ca, c.co)
            self.nops += 1
        if not need_gx:
            return 0
        gx = self.galloc(n * ca * self.esz)
        self.w += (OP_CONV_APPLY_IMAGE | branch, g, c.img_b, 0, 0, 0, gx, n, n, cb, ca, 1, 0, self.code, 0, 0, 0, 0, 0)
        self.nops += 1
        return gx

    def b_colsum(self, g, n, c, out, flag=0):
        nb = _bn_ws(n, c) + 12 * c
        self.w += (OP_COLSUM | flag, g, self.code, n, c, out, self.scratch(nb, flag), nb)
        self.nops += 1

    def b_conv_bn(self, cb, g, g_stride, n_in, n_out, tables, need_gx=True, ci=None):
        """blocks._ConvNormAct.backward: g = gradient of the block's output [n_out, co] (rows g_stride apart)."""
        c, r = cb
        x, x1, mean, invstd = self.saved[id(cb)]
        dx = self.b_bn(r, x1, n_out, mean, invstd, g, g_stride)
        if c.transposed:
            rules, table, n_gin, kflip = tables[6:8], tables[0:3], n_in, 0
        elif c.strided:
            rules, table, n_gin, kflip = tables[6:8], tables[3:6], n_in, 0
        else:
            rules, table, n_gin, kflip = tables[3:8], tables[0:3], n_in, 1
        self.b_wgrad(c, x, n_in, dx, n_out, rules, ci)
        if not need_gx:
            return 0
        return self.b_dgrad(c, dx, n_out, table, n_gin, kflip)[0]

    def b_res(self, r, g, n, k3):
        """blocks._Residual.backward: g [n, co] contiguous -> gx [n, ci]."""
        x, x1, mean1, inv1, y1, x2, mean2, inv2, out, xs, means, invs = self.saved[id(r)]
        co = r.c2.co
        self._xjoin('tail')
        gm = self.galloc(n * co * self.esz)
        part2 = parts = nb = 0
        sums2 = sumss = nparts = 0
        if _N.tail_tiles(n, self.dtype):    # (nn/functional/norm.py _tail_backward_tiles)
            nparts = int(B.lib().lidal_bn_tail_parts(n, co, self.code))
            sums2 = self.galloc(co * nparts * 8)
            sumss = self.galloc(co * nparts * 8) if r.cs is not None else 0
            self.w += (OP_ADD_RELU_BWD_BN_TILE_SUMS, out, g, gm, self.code, n, co, x2, mean2, inv2, sums2,
                       xs if sumss else 0, means if sumss else 0, invs if sumss else 0, sumss, nparts)
        elif _N.tail_sums(n):              # (nn/functional/norm.py tail_backward)
            nb = _bn_ws(n, co)
            part2 = self.galloc(nb)
            parts = self.galloc(nb) if r.cs is not None else 0
            self.w += (OP_ADD_RELU_BWD_BN_SUMS, out, g, gm, self.code, n, co, x2, mean2, inv2, part2,
                       xs if parts else 0, means if parts else 0, invs if parts else 0, parts, nb)
        else:
            self.w += (OP_ADD_RELU_BWD, out, g, gm, n * co, self.code)
        self.nops += 1
        fl = 0
        if r.cs is not None:                # the shortcut's backward: independent of the main branch until conv1's data gradient
            # (f32 with F32_BN_ALONE: the shortcut's BatchNorm backward stays on the main stream, behind the join, too)
            fl = self.fork(2) if (BRANCH_ROWS and n >= BRANCH_ROWS and (self.bf16 or not F32_BN_ALONE)) else 0
            dxs = self.b_bn(r.bs, xs, n, means, invs, gm, co, sumss, None, flag=fl, part=(parts, nb), nparts=nparts)
            g_skip = self.b_dense(r.cs, x, n, dxs, co, branch=fl)
        else:
            g_skip = gm
        dx2 = self.b_bn(r.b2, x2, n, mean2, inv2, gm, co, sums2, 0, part=(part2, nb), nparts=nparts)
        table, rules = k3[0:3], k3[3:8]
        self.b_wgrad(r.c2, y1, n, dx2, n, rules)
        p = self.ptr
        bnb = (x1, mean1, inv1, p[r.b1.w], p[r.b1.b], 1) if self.bn_sums else None
        dy1, sums = self.b_dgrad(r.c2, dx2, n, table, n, 1, 0, bnb)
        dx1 = self.b_bn(r.b1, x1, n, mean1, inv1, dy1, r.c1.co, sums)
        self.b_wgrad(r.c1, x, n, dx1, n, rules)
        if fl:
            self.join(2)
        return self.b_dgrad(r.c1, dx1, n, table, n, 1, g_skip)[0]

    def b_add(self, a, a_stride, b, b_stride, n, c):
        out = self.galloc(n * c * self.esz)
        self.w += (OP_ADD2D, a, a_stride, b, b_stride, out, c, n, c, self.code)
        self.nops += 1
        return out

    def b_devox(self, g, lvl, c):
        """DevoxelizeFunction.backward: g [P, c] -> [n_lvl, c] (ordered per-voxel sums over the contributor lists)."""
        t = self.T.pt[lvl]
        P, m = self.T.p, self.T.n[lvl]
        gin = self.galloc(m * c * self.esz)
        if t[9] is not None:                # through the cells (F.devoxelize.cells_mode)
            nb = m * 8 * c * 4
            self.w += (OP_DEVOXELIZE_BWD_CELLS, g, t[9][0], t[9][1], t[6], t[9][2], t[9][3], gin, m, c, self.code,
                       self.scratch(nb), nb)
            self.nops += 1
            return gin
        nb = _seg_ws(8 * P, m, c)
        self.w += (OP_DEVOXELIZE_BWD_SORTED, g, t[7], t[8], t[6], gin, m, c, self.code, 8 * P,
                   self.scratch(nb) if nb else 0, nb)
        self.nops += 1
        return gin

    def b_vox(self, g, lvl, c, skip):
        """VoxelizeFunction.backward: g [n_lvl, c] (+ the gradient of the forked alias) -> [P, c]."""
        t = self.T.pt[lvl]
        P, m = self.T.p, self.T.n[lvl]
        gin = self.galloc(P * c * self.esz)
        self.w += (OP_VOXELIZE_BWD, g, t[0], t[1], skip, gin, P, m, c, self.code)
        self.nops += 1
        return gin

    def b_point(self, pt, g, need_gx=True, flag=0):
        """Backward of f_point: g [P, co] -> gradient of the Linear's input [P, ci] (the residual's is g itself).
        flag: the side stream the whole branch runs on (after a fork; the caller joins before it reads the result)."""
        lin, r = pt
        P = self.T.p
        z_in, x1, mean, invstd = self.saved[id(pt)]
        dx = self.b_bn(r, x1, P, mean, invstd, g, lin.co, flag=flag)
        gx = self.b_dense(lin.conv, z_in, P, dx, lin.co, need_gx, True, branch=flag)
        self.b_colsum(dx, P, lin.co, self.slot(lin.b), flag)
        return gx

    def point_fork(self):
        return self.fork(POINT_SIDE) if (POINT_SIDE and self.bf16) else 0

    def point_join(self):
        if POINT_SIDE and POINT_SIDE in self.open:
            self.join(POINT_SIDE)

    def b_dropout(self, which, g, n, c):
        noise = self.noise[which]
        if noise is None:
            return g
        self.flush()
        gt = self.barena.tensor(g, (n, c), self.dtype)
        if isinstance(noise, int):
            gt.zero_()
        else:
            torch.mul(gt, noise, out=gt)
        return g

    def backward(self, g_logits, g_feat):
        if self.done:
            raise RuntimeError('lidal_amd.plan: this step has already run its backward pass (its activations are '
                               'released; retain_graph is not supported by the planned step)')
        if not self.geometry.alive():
            raise RuntimeError('lidal_amd: this backward pass reads the coordinate tables of a geometry that is stale -- two '
                               'newer ones have been submitted to its prefetcher since (queue forward AND backward of a '
                               'batch before the second submit() after its own: network/geometry.py GeometryPrefetcher)')
        prog, T = self.prog, self.T
        n, k3, k2 = T.n, T.k3, T.k2
        e = self.esz
        self.stream = B.stream()
        self.barena = _Arena(self.dev)
        flat = torch.empty(prog.flat_numel + 16, dtype=torch.float32, device=self.dev)
        self.flat = flat.data_ptr()
        assert self.flat % 16 == 0
        # ---- classifier
        lin = prog.classifier
        x, rows, co_pad = self.saved['cls']
        g = g_logits
        if g.dtype != self.dtype or not g.is_contiguous():
            g = g.contiguous().to(self.dtype)
        gp = self.galloc(rows * co_pad * e)
        self.w += (OP_COPY2D, g.data_ptr(), lin.co * e, gp, co_pad * e, rows, lin.co * e, (co_pad - lin.co) * e)
        self.nops += 1
        gy = self.b_dense(lin.conv, x, rows, gp, co_pad, True, True)
        self.b_colsum(gp, rows, co_pad, self.slot(lin.b))          # (its slot is the last: the padding columns fall behind it)
        c_out = self.c_feat
        if g_feat is not None:
            gf = g_feat if (g_feat.dtype == self.dtype and g_feat.is_contiguous()) else g_feat.contiguous().to(self.dtype)
            self.keep_gf = gf
            gy = self.b_add(gy, c_out, gf.data_ptr(), c_out, rows, c_out)
        gz_lin = 0
        if prog.spvcnn:
            gz_lin = self.b_point(prog.points[2], gy, flag=self.point_fork())          # gradient of z2.F through the Linear
            gy = self.b_devox(gy, 0, c_out)
        g_skip = [0] * 5        # (address, row stride) of the concatenation's gradient slice of every encoder level
        for i in (3, 2, 1, 0):
            dec, ra, rb = prog.ups[i]
            l = 3 - i
            if prog.spvcnn and i == 1:
                # y was vox(z2) with dropout, z2 = point(z1) + devox(up2 output)
                c2 = rb.c2.co
                gy = self.b_dropout(1, gy, n[2], c2)
                self.point_join()
                gz = self.b_vox(gy, 2, c2, gz_lin)
                gz_lin = self.b_point(prog.points[1], gz, flag=self.point_fork())
                gy = self.b_devox(gz, 2, c2)
            gy = self.b_res(rb, gy, n[l], k3[l])
            gcat = self.b_res(ra, gy, n[l], k3[l])
            c_up, c_cat = dec[0].co, ra.c1.ci
            g_skip[l] = (gcat + c_up * e, c_cat)
            gy = self.b_conv_bn(dec, gcat, c_cat, n[l + 1], n[l], k2[l])
        cs4 = prog.stages[3][2].c2.co
        if prog.spvcnn:
            gy = self.b_dropout(0, gy, n[4], cs4)
            self.point_join()
            gz = self.b_vox(gy, 4, cs4, gz_lin)
            gz_lin = self.b_point(prog.points[0], gz, flag=self.point_fork())              # gradient of z0.F through the Linear
            gy = self.b_devox(gz, 4, cs4)
        for l in (3, 2, 1, 0):
            down, ra, rb = prog.stages[l]
            c = rb.c2.co
            if l < 3:                       # the level also fed a decoder concatenation: autograd's sum of the two
                gy = self.b_add(gy, c, g_skip[l + 1][0], g_skip[l + 1][1], n[l + 1], c)
            gy = self.b_res(rb, gy, n[l + 1], k3[l + 1])
            gy = self.b_res(ra, gy, n[l + 1], k3[l + 1])
            gy = self.b_conv_bn(down, gy, down[0].co, n[l], n[l + 1], k2[l])
        c0 = prog.stem[1][0].co
        if prog.spvcnn:
            self.point_join()
            gz0 = self.b_vox(gy, 0, c0, gz_lin)
            gy = self.b_devox(gz0, 0, c0)
        gy = self.b_add(gy, c0, g_skip[0][0], g_skip[0][1], n[0], c0)
        gy = self.b_conv_bn(prog.stem[1], gy, c0, n[0], n[0], k3[0])
        self.b_conv_bn(prog.stem[0], gy, prog.stem[0][0].co, n[0], n[0], k3[0], False, 8 if self.bf16 else 4)
        self.flush()
        self.done = True
        self.flat_t = None
        if self.tapes is not None and TRACE is not None:         # (a test replays the plan: it releases the memory)
            self.flat_t = flat
            TRACE.append(self)
        else:
            self.release()
        return self._grads(flat)

    def release(self):
        """The activations (and the backward pass's gradients) go back to the allocator: every kernel that reads
        them is queued."""
        self.arena.release()
        if self.barena is not None:
            self.barena.release()
        self.saved = {}
        self.noise = []
        self.flat_t = None

    def _grads(self, flat):
        prog = self.prog
        views = [None] * len(prog.params)
        if prog.flat_exact:
            # one C++ call for the 161 views (split_with_sizes + a view each from Python: 0.45 ms of a host-bound step)
            order = prog.flat_order
            if prog.flat_params is None:
                prog.flat_params = [prog.params[i] for i in order]
            parts = torch._C._nn.unflatten_dense_tensors(flat[:prog.flat_numel], prog.flat_params)
            for i, t in zip(order, parts):
                views[i] = t
        else:
            for i, p in enumerate(prog.params):
                views[i] = flat[prog.slot[i]:prog.slot[i] + p.numel()].view(p.shape)
        return views


class _EvalRun(_Run):
    """One inference pass (score/prob_inference.py:97-99 under model.eval() / no_grad) as ONE plan: every
    Conv3d -> BatchNorm (-> ReLU) is one kernel with the folded BatchNorm map in its epilogue, a residual block's sum
    and final ReLU ride in its last convolution, the point-branch sum in the Linear's epilogue -- the launches of
    blocks.ConvNormSequential's inference path, same arguments, nothing saved."""
    TRAIN = False

    def __init__(self, model, prog, geometry, feats, code):
        prog.eval_operands(feats.device)
        _Run.__init__(self, model, prog, geometry, feats, code)

    def ccode(self, c):
        # f32 inference: the split form wherever the layer's reduction is whole 32-channel slices (backend.conv_code)
        return B.conv_code(self.dtype, c.ci, True)

    def _constants(self):
        prog = self.prog
        if prog.ones is None or prog.ones.device != self.dev:
            prog.ones = torch.ones(512, dtype=torch.float32, device=self.dev)
            prog.cls_shift = torch.zeros(64, dtype=torch.float32, device=self.dev)
        self.ones = prog.ones.data_ptr()
        self.cls_shift = prog.cls_shift.data_ptr()

    def f_conv_bn(self, cb, x, n_in, table, n_out, ci=None):
        c, r = cb
        out = self.arena.alloc(n_out * c.co * self.esz)
        wb = _C.apply_workspace_bytes(n_out, c.co)
        self.w += (OP_CONV_APPLY_IMAGE_WS, x, c.img_f, table[0], table[1], table[2], out, n_in, n_out,
                   c.ci if ci is None else ci, c.co, c.k, 0, self.ccode(c), r.scale, r.shift, r.relu, 0, 0,
                   self.scratch(wb) if wb else 0, wb)
        self.nops += 1
        return out

    def f_res(self, r, x, n, table):
        A, e = self.arena, self.esz
        skip = x
        fl = 0
        if r.cs is not None:                # the shortcut: one dense kernel (on a side stream it measured no gain at
            fl = 0                          # inference, and a loss beside the scorer's stream: scripts/exp/score_ab.sh)
            skip = A.alloc(n * r.cs.co * e)
            self.w += (OP_CONV_APPLY_IMAGE | fl, x, r.cs.img_f, 0, 0, 0, skip, n, n, r.cs.ci, r.cs.co, 1, 0, self.ccode(r.cs),
                       r.bs.scale, r.bs.shift, 0, 0, 0)
            self.nops += 1
        y1 = A.alloc(n * r.c1.co * e)
        out = A.alloc(n * r.c2.co * e)
        wb1, wb2 = _C.apply_workspace_bytes(n, r.c1.co), _C.apply_workspace_bytes(n, r.c2.co)
        self.w += (OP_CONV_APPLY_IMAGE_WS, x, r.c1.img_f, table[0], table[1], table[2], y1, n, n, r.c1.ci, r.c1.co, r.c1.k,
                   0, self.ccode(r.c1), r.b1.scale, r.b1.shift, 1, 0, 0, self.scratch(wb1) if wb1 else 0, wb1)
        self.nops += 1
        if fl:
            self.join(2)
        self.w += (OP_CONV_APPLY_IMAGE_WS, y1, r.c2.img_f, table[0], table[1], table[2], out, n, n, r.c2.ci, r.c2.co,
                   r.c2.k, 0, self.ccode(r.c2), r.b2.scale, r.b2.shift, 2, skip, 0, self.scratch(wb2) if wb2 else 0, wb2)
        self.nops += 1
        return out

    def f_point(self, pt, z_in, devox_out, pre=None):
        lin, r = pt
        P = self.T.p
        out = self.arena.alloc(P * lin.co * self.esz)
        shift = self.prog._point_shift[self.prog.points.index(pt)]
        self.w += (OP_CONV_APPLY_IMAGE, z_in, lin.conv.img_f, 0, 0, 0, out, P, P, lin.ci, lin.co, 1, 0, self.ccode(lin.conv),
                   r.scale, shift, 1, devox_out, 0)
        self.nops += 1
        return out

    def f_dropout(self, addr, n, c):
        return

    def forward(self):
        out = _Run.forward(self)
        self.saved = {}
        if self.tapes is not None and TRACE is not None:        # (a test replays the plan: the run keeps its arena alive)
            TRACE.append(self)
        return out


class _PlannedNet(torch.autograd.Function):
    """The whole network as one autograd node: forward = one plan, backward = one plan."""

    @staticmethod
    def forward(ctx, run, *params):
        ctx.run = run
        ctx.set_materialize_grads(False)
        return run.forward()

    @staticmethod
    def backward(ctx, g_logits, g_feat):
        B.note_backward()
        run = ctx.run
        if g_logits is None:
            rows = run.T.p if run.prog.spvcnn else run.T.n[0]
            g_logits = torch.zeros((rows, run.prog.n_class), dtype=run.dtype, device=run.dev)
        return (None,) + tuple(run.backward(g_logits, g_feat))


def planned_forward(model, x):
    """model(x) as launch plans -- a training step as one planned autograd node, an inference pass (eval mode,
    no_grad) as one plan -- or None when the configuration is not a planned one."""
    mode = plannable(model, x)
    if mode is None:
        return None
    from .geometry import Geometry
    g = getattr(x, 'geometry', None)
    if g is None:
        g = Geometry.build(model, x.C, mode == 'train')
        if g.z is None:                 # torchsparse leaves the maps it built on the input tensor's dicts (MinkUNet's
            x.cmaps.update(g.x0.cmaps)  # level 0 IS the input): visible to the caller as with the per-operator path
            x.kmaps.update(g.x0.kmaps)
    else:
        g.admit(x, type(model).__name__)
        if mode == 'train' and not g.grad:
            return None
    prog = model.__dict__['_lidal_program']
    code = B.BF16 if B.compute_dtype(x.F) == torch.bfloat16 else B.F32
    if mode == 'eval':
        return _EvalRun(model, prog, g, x.F, code).forward()
    run = _Run(model, prog, g, x.F, code)
    return _PlannedNet.apply(run, *prog.params)
