"""victim3 (forms of the packed f32 instructions) beside a loop of v_mfma_f32_16x16x32_bf16 (aggressor.hip mode 0) and beside
v_mfma_f32_32x32x16_bf16 (mode 1)."""
import ctypes, os
HERE = os.path.dirname(os.path.abspath(__file__))
import torch
dev = torch.device('cuda')
vp = ctypes.c_void_p
v3 = ctypes.CDLL(os.path.join(HERE, 'libvictim3.so'))
v3.victim3_launch.argtypes = [vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, vp]
ag = ctypes.CDLL(os.path.join(HERE, 'libaggressor.so'))
ag.aggressor_launch.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
FORMS = ['pk_mul op_sel:[0,1] op_sel_hi:[0,1]', 'pk_mul op_sel_hi:[0,1]', 'pk_mul op_sel_hi:[1,0]', 'pk_mul op_sel:[1,0] op_sel_hi:[1,0]',
         'pk_mul op_sel:[1,1] op_sel_hi:[0,0]', 'pk_add op_sel_hi:[1,0]', 'pk_add op_sel:[0,1] op_sel_hi:[1,0]', 'pk_fma op_sel_hi:[0,1,1]',
         'pk_fma op_sel:[0,1,0] op_sel_hi:[0,1,1]', 'pk_fma (plain)', 'pk_mul (plain)', 'v_fma_f64']
fsrc = torch.rand(1 << 22, device=dev) + 0.5
out = torch.empty(4096 * 256, device=dev)
report = torch.zeros(16, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream().cuda_stream
for mode, name in ((0, 'v_mfma_f32_16x16x32_bf16'), (1, 'v_mfma_f32_32x32x16_bf16'), (-1, 'nothing')):
    report.zero_()
    for it in range(10):
        if mode >= 0:
            ag.aggressor_launch(fsrc.data_ptr(), out.data_ptr(), 2048, 20000, mode, main)
        with torch.cuda.stream(side):
            assert v3.victim3_launch(fsrc.data_ptr(), fsrc.numel(), 2048, 64, report.data_ptr(), side.cuda_stream) == 0
        torch.cuda.synchronize()
    r = report.tolist()
    print('beside %s (of %d evaluations per form):' % (name, 10 * 2048 * 256 * 64))
    for k, f in enumerate(FORMS):
        print('   %-44s wrong %d' % (f, r[k]))
