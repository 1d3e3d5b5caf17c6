"""victim4 (f64 VALU forms, f32 transcendental / division forms, a 64-lane f64 butterfly sum) beside loops of
v_mfma_f32_16x16x32_bf16 (aggressor.hip mode 0), v_mfma_f32_16x16x4_f32 (mode 6), v_mfma_f32_32x32x16_bf16 (mode 1) and alone.
Build here first:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -o libvictim4.so victim4.hip ; same for aggressor.hip"""
import ctypes, os
HERE = os.path.dirname(os.path.abspath(__file__))
import torch
dev = torch.device('cuda')
vp = ctypes.c_void_p
v4 = ctypes.CDLL(os.path.join(HERE, 'libvictim4.so'))
v4.victim4_launch.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]
ag = ctypes.CDLL(os.path.join(HERE, 'libaggressor.so'))
ag.aggressor_launch.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
FORMS = ['v_fma_f64', 'v_add_f64', 'v_mul_f64', 'v_cvt_f64_f32', 'v_cvt_f32_f64', 'v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64',
         'f64 division (compiler)', 'f64 sqrt (compiler)', '64-lane f64 butterfly sum', 'v_fma_f32', 'v_rcp_f32', 'v_rsq_f32',
         'f32 division (compiler)', 'v_max_f64']
fsrc = torch.rand(1 << 22, device=dev) + 0.5
out = torch.empty(4096 * 256, device=dev)
report = torch.zeros(16, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get('REPS', '10'))
for mode, name in ((0, 'v_mfma_f32_16x16x32_bf16'), (6, 'v_mfma_f32_16x16x4_f32'), (1, 'v_mfma_f32_32x32x16_bf16'), (-1, 'nothing')):
    report.zero_()
    for it in range(REPS):
        if mode >= 0:
            ag.aggressor_launch(fsrc.data_ptr(), out.data_ptr(), 2048, 20000, mode, main)
        with torch.cuda.stream(side):
            assert v4.victim4_launch(fsrc.data_ptr(), fsrc.numel(), 2048, 64, report.data_ptr(), side.cuda_stream) == 0
        torch.cuda.synchronize()
    r = report.tolist()
    print('beside %s (of %d evaluations per form):' % (name, REPS * 2048 * 256 * 64))
    for k, f in enumerate(FORMS):
        print('   %-30s mismatches %d' % (f, r[k]))
