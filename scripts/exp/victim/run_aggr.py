"""victim2 (packed f32 multiply with op_sel) beside synthetic aggressors (aggressor.hip) and library GEMMs."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
import torch
dev = torch.device('cuda')
vp = ctypes.c_void_p
v2 = ctypes.CDLL(os.path.join(HERE, 'libvictim2.so'))
v2.victim2_launch.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]
ag = ctypes.CDLL(os.path.join(HERE, 'libaggressor.so'))
ag.aggressor_launch.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
fsrc = torch.rand(1 << 22, device=dev) + 0.5
out = torch.empty(4096 * 256, device=dev)
report = torch.zeros(8, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
a16 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
a32 = torch.randn(4096, 4096, device=dev)
main = torch.cuda.current_stream().cuda_stream
names = ['v_mfma_f32_16x16x32_bf16', 'v_mfma_f32_32x32x16_bf16', 'LDS-DMA b128', 'ds_read_b128', 'VALU fma',
         'v_mfma_f32_16x16x16_bf16', 'v_mfma_f32_16x16x4_f32']
jobs = [(n, (lambda m=m: ag.aggressor_launch(fsrc.data_ptr(), out.data_ptr(), 2048, 20000, m, main))) for m, n in enumerate(names)]
jobs += [('torch bf16 matmul 8192^3', lambda: a16 @ a16), ('torch f32 matmul 4096^3', lambda: a32 @ a32), ('nothing', lambda: None)]
for name, job in jobs:
    job(); torch.cuda.synchronize()
    report.zero_()
    for it in range(10):
        job()
        with torch.cuda.stream(side):
            assert v2.victim2_launch(fsrc.data_ptr(), fsrc.numel(), 2048, 64, report.data_ptr(), side.cuda_stream) == 0
        torch.cuda.synchronize()
    r = report.tolist()
    print('%-28s: wrong v_pk_mul_f32 op_sel %d, v_pk_mul_f32 %d, v_mul_f32 %d, v_pk_add_f32 neg %d' % (name, r[0], r[1], r[2], r[3]), flush=True)
