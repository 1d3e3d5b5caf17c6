"""Dense per-row products on [N, C] feature matrices: the 1x1x1 convolutions (F.conv3d's first
branch, `feats.matmul(weight)`) and the point-branch nn.Linear layers (network/spvcnn.py:85-101).

Forward and data gradient run on the sparse convolution kernel with the identity rule list (bf16 and
f32 alike: exact f32 MFMA in the parity mode -- no library GEMM on the path).  The WEIGHT gradient
x^T [Cin, N] @ g [N, Cout] reduces over N ~ 4e5 rows into a tiny [Cin, Cout] tile: a library GEMM
runs that in a few workgroups (0.8 ms measured for 128x96), so it goes through the split-K MFMA
kernel used for the sparse weight gradients (lidal_conv_wgrad with the identity rule list)."""
import torch
from torch.autograd import Function

from ... import backend as B
from .conv import wgrad_scratch

__all__ = ['rows_matmul', 'rows_linear', 'rows_backward']

_koff_cache = {}


def _koff(n, device):
    key = (n, str(device))
    t = _koff_cache.get(key)
    if t is None:
        if len(_koff_cache) > 256:
            _koff_cache.clear()
        t = torch.tensor([0, n], dtype=torch.int64, device=device)
        _koff_cache[key] = t
    return t


def _wgrad_dense(a, b):
    """a [N, Ca], b [N, Cb] (same dtype, f32 or bf16) -> a^T @ b as f32 [Ca, Cb]."""
    n, ca = a.shape
    cb = b.shape[1]
    gw = torch.empty((1, ca, cb), dtype=torch.float32, device=a.device)
    from .conv import wgrad_plan
    code, slabs = wgrad_plan(n, n, 1, ca, cb, a.dtype)          # (f32: the split form, backend.wgrad_code)
    partial = torch.empty((slabs, ca, cb), dtype=torch.float32, device=a.device)
    B.check(B.lib().lidal_conv_wgrad(B.ptr(a), B.ptr(b), n, n, None, B.ptr(_koff(n, a.device)), 0,
                                     B.ptr(gw), B.ptr(partial), partial.shape[0], 1, ca, cb,
                                     code, B.stream()), 'conv_wgrad(dense)')
    return gw[0]


def _ok(x, ca, cb):
    vec = 8 if x.dtype == torch.bfloat16 else 4
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and ca % vec == 0
            and cb % vec == 0 and x.shape[0] > 0)


def _vec(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _gemm_ok(x, ci, co):
    if not x.is_cuda or x.shape[0] == 0 or x.shape[0] * ci * x.element_size() >= 0x7FFFFFF0:
        return False
    vec = _vec(x.dtype) if x.dtype in (torch.float32, torch.bfloat16) else 0
    return vec > 0 and ci % vec == 0 and co % vec == 0


def _rows_gemm(x, w, role, shift=None, scale=None, relu=False, residual=None, img=None, want_stats=False, code=None):
    """x [n, n_red] times the [Cin, Cout] operand `w` (compute dtype): role 0 = x @ w (reduction over
    Cin), role 1 = x @ w^T (reduction over Cout: the data gradient), + shift f32: the sparse
    convolution kernel with the identity rule list (a NULL table) and the weight as an LDS image.
    A library GEMM runs these tall-skinny products (4e5 x 128 @ 128 x 96) at ~1.7 TB/s of operand
    traffic; this kernel streams the rows once and keeps the small weight in LDS."""
    from .conv import _weight_image
    n, n_red = x.shape
    n_col = w.shape[1] if role == 0 else w.shape[0]
    assert n_red == (w.shape[0] if role == 0 else w.shape[1])
    if img is None:
        img = _weight_image(w.contiguous().unsqueeze(0), x.dtype, n, role, code)
    out = torch.empty((n, n_col), dtype=x.dtype, device=x.device)
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == out.shape and residual.dtype == out.dtype
    if shift is not None and scale is None:
        scale = torch.ones(n_col, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        stats = torch.empty((-(-n // B.stats_tile_rows()), n_col, 3), dtype=torch.float32, device=x.device)
    B.check(B.lib().lidal_conv_apply_image(B.ptr(x), B.ptr(img), None, None, None, B.ptr(out),
                                           n, n, n_red, n_col, 1, 0, B.dtype_code(x.dtype) if code is None else code, B.ptr(scale),
                                           B.ptr(shift), int(relu), B.ptr(residual), B.ptr(stats),
                                           B.stream()),
            'conv_apply(dense)')
    if want_stats:
        out._lidal_bn_stats = stats
    return out


def _operand(w, linear, cdtype, pad):
    """The GEMM operand [Cin, Cout+pad] in the compute dtype for `w` ([Cin, Cout], or nn.Linear's
    [Cout, Cin] when `linear`)."""
    wc = w.detach().to(cdtype)
    if linear:
        wc = wc.t()
    if pad:
        wc = torch.nn.functional.pad(wc, (0, pad))
    return wc


def _forward(x, w, bias, linear, epilogue=None, with_bwd_image=False, want_stats=False, inference=False):
    """epilogue (inference only) = (scale f32 [Cout], shift f32 [Cout], relu[, residual [N, Cout]]):
    the eval-mode BatchNorm (+ ReLU) that follows the layer and an optional row-wise sum,
    y = act((x @ w + bias) * scale + shift) + residual."""
    img_b = None
    cdtype = B.compute_dtype(x)
    xc = x.contiguous().to(cdtype)
    co = w.shape[0] if linear else w.shape[1]
    pad = (-co) % _vec(cdtype) if xc.is_cuda else 0
    from . import conv as C
    # training, a plain f32 parameter: its images come from the step's one batched launch, built from
    # the parameter itself -- no cast / transposed copy of the operand is needed (shape only)
    banked = (with_bwd_image and epilogue is None and C._IMAGE_BATCH and not pad
              and isinstance(w, torch.nn.Parameter) and w.is_contiguous() and w.dtype == torch.float32
              and _gemm_ok(xc, xc.shape[1], co))
    wc = (torch.empty((xc.shape[1], co), dtype=cdtype, device='meta') if banked
          else _operand(w, linear, cdtype, pad))
    scale = shift = residual = None
    relu = False
    if epilogue is not None:
        scale, shift, relu = epilogue[:3]
        residual = epilogue[3] if len(epilogue) > 3 else None
        if bias is not None:
            shift = shift + bias.detach().float() * scale
    elif bias is not None:
        shift = bias.detach().float()
    if _gemm_ok(xc, xc.shape[1], co + pad):
        if pad:
            shift = None if shift is None else torch.nn.functional.pad(shift, (0, pad))
            scale = None if scale is None else torch.nn.functional.pad(scale, (0, pad), value=1.0)
        fused_res = residual if (residual is not None and not pad) else None
        if fused_res is not None:
            fused_res = fused_res.to(cdtype)
        late = residual is not None and fused_res is None       # sum (and its ReLU) outside the kernel
        img_f = None
        img_key = None
        # f32 inference: the split form (backend.conv_code) -- only where no autograd node will be made
        code = B.conv_code(cdtype, xc.shape[1], inference and not with_bwd_image and not want_stats)
        # (grad mode is also off inside RowsMatmul.forward: `with_bwd_image` tells training apart)
        if not with_bwd_image and not torch.is_grad_enabled():     # inference: the image of an unchanged parameter is re-used
            img_key = (B.weights_key(w), linear, code, pad,
                       B.lib().lidal_conv_weight_image_tiling(xc.shape[1], co + pad, code, xc.shape[0]))
            cache = getattr(w, '_lidal_images', None)
            if cache is not None and img_key in cache:
                img_f = cache[img_key]
            else:
                from .conv import _weight_image
                with torch.enable_grad():       # (bypass the per-tensor cache of the temporary operand)
                    img_f = _weight_image(wc.contiguous().unsqueeze(0), cdtype, xc.shape[0], 0, code)
                if cache is None or next(iter(cache))[0] != B.weights_key(w):
                    cache = {}
                    w._lidal_images = cache
                cache[img_key] = img_f
        if with_bwd_image:      # forward and data-gradient operands of this parameter from one launch
            if banked:          # ... straight from the f32 parameter in its own layout
                img_f, img_b = C._IMAGE_BANK.get(w, cdtype, xc.shape[0], xc.shape[0],
                                                 (1, xc.shape[1], co), 1 if linear else 0)
            else:
                img_f, img_b = C._weight_image_pair(wc.contiguous().unsqueeze(0), cdtype, xc.shape[0], xc.shape[0])
        want_stats = want_stats and not pad and not late and cdtype == torch.bfloat16
        y = _rows_gemm(xc, wc, 0, shift, scale,
                       int(relu) & 1 if late else int(relu), fused_res, img_f, want_stats, code)
        y = y[:, :co] if pad else y
        if late:
            y = y + residual.to(cdtype)
            y = torch.relu(y) if int(relu) & 2 else y
        return xc, wc, pad, y, img_b
    B.hit('library_gemm:rows')        # a shape the kernel does not take (channels not a multiple of 16 bytes)
    y = xc @ wc
    y = y[:, :co] if pad else y
    if epilogue is not None:
        y = y.float() * scale + shift
        y = torch.relu(y) if int(relu) & 1 else y
        if residual is not None:
            y = y + residual.float()
            y = torch.relu(y) if int(relu) & 2 else y
        y = y.to(cdtype)
    elif bias is not None:
        y = y + bias.detach().to(cdtype)
    return xc, wc, pad, y, img_b


class RowsMatmul(Function):
    """y = x @ w  (+ bias), w [Cin, Cout] (or nn.Linear's [Cout, Cin] with linear=True).  A Cout
    that is not a multiple of the 16-byte vector width (the 19-class classifier) is zero-padded
    internally so that forward, weight gradient and bias gradient all stay on the vectorised
    kernels; the caller sees exactly [N, Cout]."""
    last_stats = None

    @staticmethod
    def forward(ctx, x, w, bias, linear, want_stats=False):
        xc, wc, pad, y, ctx.img_bwd = _forward(x, w, bias, linear, None, ctx.needs_input_grad[0], want_stats)
        RowsMatmul.last_stats = getattr(y, '_lidal_bn_stats', None)
        ctx.save_for_backward(xc, w)
        ctx.wc = wc                      # operand in the compute dtype, re-used by the data gradient
        ctx.has_bias = bias is not None
        ctx.pad = pad
        ctx.linear = linear
        return y

    @staticmethod
    def backward(ctx, g):
        B.note_backward()
        xc, w = ctx.saved_tensors
        return rows_backward(xc, w, ctx.wc, ctx.img_bwd, ctx.pad, ctx.linear, g, ctx.needs_input_grad[0],
                             ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]) + (None, None)


def rows_backward(xc, w, wc, img_bwd, pad, linear, g, need_gx=True, need_gw=True, need_gb=False, grad_skip=None):
    """Backward of y = x @ w on raw tensors (xc, wc, img_bwd, pad as _forward returned them):
    -> (gx, gw, gb).  `grad_skip` ([N, Cin], the compute dtype): added to gx in the kernel's epilogue.
    Shared by RowsMatmul and the fused block Functions of lidal_amd.network."""
    g = g.to(xc.dtype)
    co = w.shape[0] if linear else w.shape[1]
    if pad:
        g = torch.nn.functional.pad(g, (0, pad))
    g = g.contiguous()
    gx = gw = gb = None

    def wgrad():
        if _ok(xc, xc.shape[1], g.shape[1]):
            gw_ = _wgrad_dense(xc, g)[:, :co]
        else:
            gw_ = (xc.float().t() @ g.float())[:, :co]
        return (gw_.t().contiguous() if linear else gw_.contiguous()).to(w.dtype)

    side = None
    if need_gw:
        if B.overlap_wgrad(xc.dtype) and need_gx and g.is_cuda:
            side = B.beside(g.device, (xc, g), w)
            with side as done:              # beside the data gradient below (backend.beside)
                gw = wgrad()
                done(gw)
        else:
            gw = wgrad()
    if need_gx:
        if _gemm_ok(g, g.shape[1], xc.shape[1]):
            gx = _rows_gemm(g, wc, 1, residual=grad_skip, img=img_bwd)             # reduction over co
        else:
            gx = g @ wc.t()
            if grad_skip is not None:
                gx = gx + grad_skip
    if side is not None:
        side.finish()
    if need_gb:
        from .norm import column_sum
        if g.is_cuda and g.shape[1] % _vec(g.dtype) == 0 and g.shape[1] // _vec(g.dtype) <= 256:
            gb = column_sum(g)[:co]
        else:
            gb = g.float().sum(0)[:co]
    return gx, gw, gb


def _rows(x, w, bias, linear, epilogue=None, want_stats=False):
    if B.wants_grad(x, w, bias):
        assert epilogue is None, 'the fused BatchNorm epilogue is inference-only'
        y = RowsMatmul.apply(x, w, bias, linear, want_stats)
        if want_stats and RowsMatmul.last_stats is not None:
            y._lidal_bn_stats = RowsMatmul.last_stats
        RowsMatmul.last_stats = None
        return y
    return _forward(x, w, bias, linear, epilogue, False, want_stats, True)[3]    # inference: no autograd node


def rows_matmul(x, w, bias=None, epilogue=None, want_stats=False):
    return _rows(x, w, bias, False, epilogue, want_stats)


def rows_linear(x, weight, bias=None, epilogue=None, want_stats=False):
    """nn.Linear semantics: weight [Cout, Cin].  `want_stats`: leave the batch statistics of a
    train-mode BatchNorm that follows on the output (see conv.conv3d)."""
    return _rows(x, weight, bias, True, epilogue, want_stats)
