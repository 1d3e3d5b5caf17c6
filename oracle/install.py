"""Register the CPU oracle under the module name `torchsparse` so that the reference's own
network/*.py can be imported UNCHANGED in the build container (tests/golden/make_golden.py).
TEST INFRASTRUCTURE; never imported by lidal_amd/."""
import sys


def install_as_torchsparse():
    from oracle import tsref
    from oracle.tsref import nn as tsnn
    from oracle.tsref.nn import functional as tsF
    from oracle.tsref.nn import utils as tsU
    sys.modules['torchsparse'] = tsref
    sys.modules['torchsparse.nn'] = tsnn
    sys.modules['torchsparse.nn.functional'] = tsF
    sys.modules['torchsparse.nn.utils'] = tsU
    tsref.nn = tsnn
    tsnn.functional = tsF
    tsnn.utils = tsU
    return tsref
