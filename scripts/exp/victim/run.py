"""Runs victim.hip's kernel on a second stream beside a convolution on the main stream; prints what changed under the
victim: VGPR patterns, SGPR patterns, LDS, global loads of a known buffer, a compare->mask->select."""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import torch
from lidal_amd import SparseTensor, synth
from lidal_amd import nn as spnn
from lidal_amd.network import SPVCNN, Geometry

dev = torch.device('cuda')
lib = ctypes.CDLL(os.path.join(HERE, 'libvictim.so'))
lib.victim_launch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
n = 1 << 22
src = (torch.arange(n, device=dev, dtype=torch.int64) * 2654435761 % (1 << 32)).to(torch.int64)
src = (src & 0xFFFFFFFF).to(torch.uint32) if hasattr(torch, 'uint32') else None
report = torch.zeros(8, dtype=torch.int32, device=dev)
b2 = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
coords2 = torch.from_numpy(b2['coords_v_b']).to(dev)
model = SPVCNN(19).to(dev).train()
g2 = Geometry.build(model, coords2, grad=False)


def level(stride, c):
    st = (stride,) * 3
    cs = g2.x0.cmaps[st]
    x = SparseTensor(torch.randn(cs.shape[0], c, device=dev).bfloat16(), cs, stride)
    x.cmaps, x.kmaps = g2.x0.cmaps, g2.x0.kmaps
    return x


def conv_job(stride, ci, co, k=3):
    conv = spnn.Conv3d(ci, co, k).to(dev)
    x = level(stride, ci)

    def run():
        with torch.autocast('cuda', dtype=torch.bfloat16), torch.no_grad():
            conv(x)
    return run


lib2 = ctypes.CDLL(os.path.join(HERE, 'libvictim2.so'))
lib2.victim2_launch.argtypes = lib.victim_launch.argtypes
fsrc = torch.rand(1 << 22, device=dev) + 0.5
side = torch.cuda.Stream()
jobs = {'nothing': lambda: None, 'dense 96->96': conv_job(1, 96, 96, 1), 'conv 32->32 k3': conv_job(1, 32, 32),
        'conv 96->96 k3': conv_job(1, 96, 96), 'conv 256->256 k3 s8': conv_job(8, 256, 256)}
for name, job in jobs.items():
    for _ in range(3):
        job()
    torch.cuda.synchronize()
    report.zero_()
    for it in range(30):
        for _ in range(6):
            job()
        with torch.cuda.stream(side):
            rc = lib.victim_launch(src.data_ptr(), n, 2048, 200, report.data_ptr(), side.cuda_stream)
            assert rc == 0
        torch.cuda.synchronize()
    r = report.tolist()
    print('%-22s: changed VGPRs %d, SGPRs %d, LDS words %d, wrong global loads %d, wrong selects %d' % (name, r[0], r[1], r[2], r[3], r[4]), flush=True)
    report.zero_()
    for it in range(30):
        for _ in range(6):
            job()
        with torch.cuda.stream(side):
            assert lib2.victim2_launch(fsrc.data_ptr(), fsrc.numel(), 2048, 64, report.data_ptr(), side.cuda_stream) == 0
        torch.cuda.synchronize()
    r = report.tolist()
    print('%-22s: wrong v_pk_mul_f32 op_sel %d, v_pk_mul_f32 %d, v_mul_f32 %d, v_pk_add_f32 neg %d, cmp/select %d' % ('', r[0], r[1], r[2], r[3], r[4]), flush=True)
