"""A/B builds of the library: recompile chosen sources with extra -D flags, link with the stock
objects into scripts/_abl/lib_<name>.so (select it with LIDAL_AMD_LIB=...).
  python scripts/build_variant.py NAME conv_img.hip -DLIDAL_IMG_G=2 [more sources / flags]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidal_amd import build as LB  # noqa: E402

OUT = os.path.join(ROOT, 'scripts', '_abl')


def main():
    name = sys.argv[1]
    srcs = [a for a in sys.argv[2:] if not a.startswith('-')]
    flags = [a for a in sys.argv[2:] if a.startswith('-')]
    LB.build(verbose=False)
    os.makedirs(os.path.join(OUT, 'obj_' + name), exist_ok=True)
    objs = []
    for s in LB.SOURCES:
        if s in srcs:
            obj = os.path.join(OUT, 'obj_' + name, s + '.o')
            cmd = ['hipcc'] + LB.FLAGS + LB.NO_PACKED_F32 + flags + (['-ffp-contract=off'] if s in LB.NO_CONTRACT else []) + \
                  (['-x', 'hip'] if s.endswith('.cpp') else []) + ['-c', os.path.join(LB.CSRC, s), '-o', obj]
            subprocess.run(cmd, check=True)
        else:
            obj = os.path.join(LB.OBJ, s + '.o')
        objs.append(obj)
    lib = os.path.join(OUT, 'lib_%s.so' % name)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, check=True)
    print('built', lib)


if __name__ == '__main__':
    main()
