"""GPU: is the f32 MFMA accumulation unbiased?  Positive operands, long reductions, signed error vs f64."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd.nn.functional.dense import _rows_gemm, _wgrad_dense  # noqa: E402

torch.manual_seed(0)
dev = 'cuda'
for n in (4096, 65536, 400000):
    a = torch.rand(n, 96, device=dev) + 0.5
    b = torch.rand(n, 96, device=dev) + 0.5
    got = _wgrad_dense(a, b).double()
    ref = a.double().t() @ b.double()
    tor = (a.t() @ b).double()
    e = ((got - ref) / ref)
    et = ((tor - ref) / ref)
    print('wgrad n=%7d  hip: mean rel err %+.2e  max |.| %.2e   torch f32 gemm: mean %+.2e max %.2e'
          % (n, e.mean().item(), e.abs().max().item(), et.mean().item(), et.abs().max().item()))
for ci in (32, 96, 384):
    x = torch.rand(100000, ci, device=dev) + 0.5
    w = torch.rand(ci, 96, device=dev) + 0.5
    got = _rows_gemm(x, w, 0).double()
    ref = x.double() @ w.double()
    tor = (x @ w).double()
    e = ((got - ref) / ref)
    et = ((tor - ref) / ref)
    print('fwd ci=%3d       hip: mean rel err %+.2e  max |.| %.2e   torch f32 gemm: mean %+.2e max %.2e'
          % (ci, e.mean().item(), e.abs().max().item(), et.mean().item(), et.abs().max().item()))
