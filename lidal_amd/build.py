"""Build liblidal_amd.so (the C-ABI HIP backend) in-tree with hipcc for gfx950.

  python -m lidal_amd.build          # compile what is out of date, link lidal_amd/liblidal_amd.so

hipcc cross-compiles without a GPU, so this also runs in the build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'liblidal_amd.so')
SOURCES = ['error.cpp', 'hash.hip', 'kmap.hip', 'voxel.hip', 'conv.hip', 'conv_img.hip', 'wgrad_dma.hip', 'wgrad_streams.hip', 'sort.hip', 'bn.hip', 'elementwise.hip', 'score.hip', 'plan.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result']
# No packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) in this library.  Measured on MI355X
# (scripts/exp/victim/, profiles/README.md "A packed multiply beside v_mfma_f32_16x16x32_bf16"): while a wave of ANOTHER
# kernel executes v_mfma_f32_16x16x32_bf16 on the same SIMD, v_pk_mul_f32 / v_pk_fma_f32 whose low result takes the HIGH
# half of a source (op_sel:[0,1]) sporadically return a wrong low result.  Kernels of one stream never overlap, but the
# coordinate tables may be built on a second stream beside the bf16 convolutions (network/geometry.py): hipcc had put
# exactly that form into ti_weights_kernel, and ~1 % of the points got a zero trilinear weight.
NO_PACKED_F32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
# units that restate numpy arithmetic (separately rounded products and sums): no fma contraction
NO_CONTRACT = {'score.hip', 'kmap.hip'}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src + '.o')
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, 'common.h'),
            os.path.join(HERE, '..', 'include', 'lidal_amd.h'), os.path.abspath(__file__)]
    if _stale(obj, deps):
        cmd = ['hipcc'] + FLAGS + NO_PACKED_F32 + (['-ffp-contract=off'] if src in NO_CONTRACT else []) + \
              (['-x', 'hip'] if src.endswith('.cpp') else []) + \
              ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    return obj


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build()
    sys.exit(0)
