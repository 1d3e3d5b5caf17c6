"""Build tests/native/liblidal_gen1.so: the first-generation convolution kernel as a TEST-ONLY shared object (it
left liblidal_amd.so in round 4).  hipcc cross-compiles for gfx950 without a GPU; __graft_entry__.build() calls
this so that the object travels to the GPU box with the snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, 'liblidal_gen1.so')


def build(verbose=True):
    sys.path.insert(0, ROOT)
    from lidal_amd import build as LB
    src = os.path.join(HERE, 'conv_gen1.hip')
    deps = [src, os.path.join(HERE, 'conv_gen1.h'), os.path.join(LB.CSRC, 'common.h'), os.path.abspath(__file__)]
    if LB._stale(LIB, deps):
        err = os.path.join(LB.CSRC, 'error.cpp')           # lidal::set_error / lidal_last_error of its own
        cmd = ['hipcc'] + LB.FLAGS + LB.NO_PACKED_F32 + ['-shared', '-x', 'hip', src, err, '-o', LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for conv_gen1.hip:\n%s' % r.stderr[-4000:])
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build()
