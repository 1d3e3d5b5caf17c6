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
SOURCES = ['error.cpp', 'hash.hip', 'kmap.hip', 'voxel.hip', 'conv.hip', 'conv_img.hip', 'wgrad_dma.hip', 'sort.hip', 'bn.hip', 'elementwise.hip', 'score.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result']
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
        cmd = ['hipcc'] + FLAGS + (['-ffp-contract=off'] if src in NO_CONTRACT else []) + \
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
