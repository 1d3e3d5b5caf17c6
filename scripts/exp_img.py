"""GPU experiment: lidal_conv_apply (v1, conv.hip) against lidal_conv_apply_image (conv_img.hip) on
the U-Net's layer shapes of the bench batch -- bitwise equality and time -- and, for the roofline
layer, under the row-order variants of scripts/exp_order.py.
  python scripts/exp_img.py            # shapes sweep
  python scripts/exp_img.py order      # + row-order variants on 96->96
Env: LIDAL_AMD_LIB selects an A/B build of the library (lidal_amd/backend.py)."""
import os
import sys

import numpy as np
import torch


def _gen1():
    """tests/native/liblidal_gen1.so: the first-generation kernel left the product library in round 4."""
    import ctypes, os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'native', 'liblidal_gen1.so'))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.lidal_conv_apply.restype = i32
    lib.lidal_conv_apply.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    return lib


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402

SHAPES = [(1, 32, 32), (1, 96, 96), (1, 128, 96), (2, 32, 32), (4, 64, 64), (4, 128, 128), (8, 256, 256),
          (8, 384, 256), (16, 256, 256)]


def timeit(fn, reps=10, rounds=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(ts)


def make_image(w_kio, dtype, n_out, role=0):
    """w_kio: [K, ci, co] f32 (torchsparse layout) -> image for forward (role 0)."""
    k, ci, co = w_kio.shape
    n_red, n_col = (ci, co) if role == 0 else (co, ci)
    nbytes = B.lib().lidal_conv_weight_image_bytes(k, n_red, n_col, B.dtype_code(dtype), n_out)
    img = torch.empty(nbytes, dtype=torch.uint8, device=w_kio.device)
    B.check(B.lib().lidal_conv_weight_image(B.ptr(w_kio), B.dtype_code(w_kio.dtype), role, B.ptr(img),
                                            B.dtype_code(dtype), k, n_red, n_col, n_out, B.stream()), 'image')
    return img


def run_pair(x, w_kio, tab, prm, tmk, n, dtype, k=27):
    ci, co = w_kio.shape[1], w_kio.shape[2]
    wt = w_kio.permute(0, 2, 1).contiguous().to(dtype)                # [K, co, ci]
    img = make_image(w_kio, dtype, n)
    o1 = torch.empty((n, co), dtype=dtype, device=x.device)
    o2 = torch.empty((n, co), dtype=dtype, device=x.device)

    def v1():
        B.check(_gen1().lidal_conv_apply(B.ptr(x), B.ptr(wt), B.ptr(tab), B.ptr(prm), B.ptr(tmk), B.ptr(o1),
                                         n, n, ci, co, k, 0, B.dtype_code(dtype), None, None, 0, None,
                                         B.stream()), 'v1')

    def v2():
        B.check(B.lib().lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(tab), B.ptr(prm), B.ptr(tmk),
                                               B.ptr(o2), n, n, ci, co, k, 0, B.dtype_code(dtype), None, None,
                                               0, None, None, B.stream()), 'v2')
    v1(), v2()
    torch.cuda.synchronize()
    same = torch.equal(o1, o2)
    err = (o1.float() - o2.float()).abs().max().item() if not same else 0.0
    return timeit(v1), timeit(v2), same, err


def main():
    dev = torch.device('cuda')
    dtype = torch.bfloat16 if os.environ.get('ABL_DTYPE', 'bf16') == 'bf16' else torch.float32
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    print('lib', B.LIB_PATH, 'dtype', dtype)
    print('%-30s %10s %10s  %s' % ('shape (rows, rules)', 'v1 us', 'image us', 'bit-equal'))
    only = os.environ.get('EXP_SHAPES')
    for stride, ci, co in SHAPES:
        if only and '%d:%d:%d' % (stride, ci, co) not in only.split(','):
            continue
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n = c.shape[0]
        g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
        x = torch.randn(n, ci, generator=g).to(dev).to(dtype)
        w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
        o = kmap.order_out
        t1, t2, same, err = run_pair(x, w, o.table, o.perm, o.tile_masks, n, dtype)
        print('s%-2d %3d->%-3d (%4dk,%5dk)      %10.1f %10.1f  %s %s'
              % (stride, ci, co, n // 1000, kmap.total // 1000, t1, t2, same, '' if same else 'max err %g' % err),
              flush=True)
    # dense form (identity rule list) and a transposed-style 8-offset map
    n = coords.shape[0]
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(n, 128, generator=g).to(dev).to(dtype)
    w = (torch.randn(1, 128, 96, generator=g) * 0.05).to(dev)
    t1, t2, same, err = run_pair(x, w, None, None, None, n, dtype, k=1)
    print('dense 128->96 (%dk rows)          %10.1f %10.1f  %s %s' % (n // 1000, t1, t2, same, err), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'order':
        import exp_order as E
        kmap, _ = F.build_kernel_map(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
        nbr = kmap.nbr_out.cpu().numpy()
        gk = E.gray_keys(nbr)
        variants = {'global': np.argsort(gk, kind='stable')}
        for blk in (4096, 8192, 16384):
            variants['block%d' % blk] = np.lexsort((gk, np.arange(n) // blk))
        g = torch.Generator(device='cpu').manual_seed(9)
        x = torch.randn(n, 96, generator=g).to(dev).to(dtype)
        w = (torch.randn(27, 96, 96, generator=g) * 0.05).to(dev)
        print('--- 96->96, row-order variants (v1 us, image us)')
        for name, perm in variants.items():
            for xcd in (False, True):
                p, t, m, act = E.tables(nbr, perm, xcd)
                pd, td, md = (torch.from_numpy(a).to(dev) for a in (p, t, m))
                t1, t2, same, err = run_pair(x, w, td, pd, md, n, dtype)
                print('%-12s xcd=%d act/tile %5.2f   %8.1f %8.1f  %s' % (name, xcd, act, t1, t2, same), flush=True)


if __name__ == '__main__':
    main()
