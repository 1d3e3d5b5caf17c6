"""lidal_amd -- MI355X-native (gfx950) backend of the LiDAL sparse-voxel hot path.

The package exposes the torchsparse 1.4.0 operator API that /root/reference/network/*.py is
written against (SparseTensor, PointTensor, cat, nn.Conv3d/BatchNorm/ReLU, nn.functional.sp*,
nn.utils.get_kernel_offsets); underneath every operator is a hand-written HIP kernel reached
through the C-ABI library liblidal_amd.so (include/lidal_amd.h).

    import lidal_amd
    lidal_amd.install_as_torchsparse()      # `import torchsparse` now resolves to this package
"""
import sys

from .operators import cat
from .tensor import PointTensor, SparseTensor

__all__ = ['SparseTensor', 'PointTensor', 'cat', 'install_as_torchsparse']
__version__ = '0.1.0'


def install_as_torchsparse():
    """Alias this package as `torchsparse` (+ .nn, .nn.functional, .nn.utils) in sys.modules so
    the reference's network/spvcnn.py, network/minkunet.py, network/utils.py, train.py and
    score/prob_inference.py import it unchanged."""
    from . import nn
    from .nn import functional, utils
    me = sys.modules[__name__]
    sys.modules['torchsparse'] = me
    sys.modules['torchsparse.nn'] = nn
    sys.modules['torchsparse.nn.functional'] = functional
    sys.modules['torchsparse.nn.utils'] = utils
    return me
