"""Verdict round 2, item 3 (ii): apply the producer's BatchNorm + ReLU to the A fragments of the consumer convolution
instead of materialising y = relu(bn(x)).  Gate: the fused convolution must beat apply + convolution on the
96->96 (level 0) and 256->256 (stride 8) layers.
  stock library :  python scripts/exp/affine_a.py                 -> time of the apply pass and of the convolution on y
  variant       :  LIDAL_AMD_LIB=scripts/_abl/lib_affine.so python scripts/exp/affine_a.py   (scripts/exp/affine_a.patch
                   applied to csrc/conv_img.hip, built with scripts/build_variant.py affine conv_img.hip -DLIDAL_EXP_AFFINE_A)
                   -> time of the convolution that reads x and applies scale / shift / ReLU in registers
Both print a checksum of the output: the two paths must agree bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidal_amd import backend as B, synth
from lidal_amd.network import SPVCNN, Geometry
from lidal_amd.nn.functional import conv as C
from exp_img import timeit

dev = torch.device('cuda')
variant = 'affine' in os.path.basename(B.LIB_PATH)
b = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords = torch.from_numpy(b['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).eval()
with torch.no_grad():
    g = Geometry.build(model, coords, grad=False)
print('library', B.LIB_PATH)
for stride, c in ((1, 96), (8, 256), (4, 128), (2, 32)):
    st = (stride,) * 3
    kmap = g.x0.kmaps[(st, (3, 3, 3), (1, 1, 1), (1, 1, 1))]
    n = g.x0.cmaps[st].shape[0]
    torch.manual_seed(stride)
    x = torch.randn(n, c, device=dev).bfloat16()
    w = (torch.randn(27, c, c, device=dev) * 0.05)
    scale = torch.rand(c, device=dev) + 0.5
    shift = torch.randn(c, device=dev) * 0.3
    y = torch.empty_like(x)
    zeros, ones = torch.zeros(c, device=dev), torch.ones(c, device=dev)

    def apply():        # y = relu(x * scale + shift): the eval-form BatchNorm kernel with (mean 0, var 1 - eps)
        B.check(B.lib().lidal_bn_eval_fwd(B.ptr(x), B.dtype_code(x.dtype), n, c, B.ptr(scale), B.ptr(shift), B.ptr(zeros),
                                          B.ptr(ones), 0.0, 1, B.ptr(y), B.stream()), 'bn_eval_fwd')
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        if not variant:
            apply()
            out = C._forward(y, w, kmap, False)[1]
            t_apply = timeit(apply)
            t_conv = timeit(lambda: C._forward(y, w, kmap, False))
            print('stride %2d  %6d rows  %3d -> %3d:  apply %6.1f us + conv %6.1f us = %6.1f   checksum %d'
                  % (stride, n, c, c, t_apply, t_conv, t_apply + t_conv, int(out.view(torch.int16).long().sum())))
        else:
            out = C._forward(x, w, kmap, False, (scale, shift, 0))[1]
            t_fused = timeit(lambda: C._forward(x, w, kmap, False, (scale, shift, 0)))
            t_plain = timeit(lambda: C._forward(x, w, kmap, False))
            print('stride %2d  %6d rows  %3d -> %3d:  conv with the affine map on its A fragments %6.1f us (same build, plain: %6.1f)   checksum %d'
                  % (stride, n, c, c, t_fused, t_plain, int(out.view(torch.int16).long().sum())))
