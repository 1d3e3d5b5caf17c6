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

__all__ = ['SparseTensor', 'PointTensor', 'cat', 'install_as_torchsparse', 'adopt_torch_modules']
__version__ = '0.1.0'


def install_as_torchsparse(adopt_torch_modules=None):
    """Alias this package as `torchsparse` (+ .nn, .nn.functional, .nn.utils) in sys.modules so
    the reference's network/spvcnn.py, network/minkunet.py, network/utils.py, train.py and
    score/prob_inference.py import it unchanged.

    adopt_torch_modules (default: on; False or LIDAL_AUTO_ADOPT=0 opts out): a model built afterwards from this
    package's Conv3d modules has its plain-torch row-wise modules (exact-type nn.Linear / nn.BatchNorm1d, and
    nn.Sequential(Linear, BatchNorm1d, ReLU) groups -- SPVCNN's point branch and classifier, network/spvcnn.py:60-98)
    handed to this package at its FIRST forward call, exactly as the explicit `adopt_torch_modules(model)` does: same
    Parameter objects, same buffers, same state_dict keys, nothing in the user's script changes (round 6; the drop-in
    step 24.6 -> 18.8 ms at 5 scans)."""
    import os
    from . import nn
    from .nn import functional, utils
    me = sys.modules[__name__]
    sys.modules['torchsparse'] = me
    sys.modules['torchsparse.nn'] = nn
    sys.modules['torchsparse.nn.functional'] = functional
    sys.modules['torchsparse.nn.utils'] = utils
    if adopt_torch_modules is None:
        adopt_torch_modules = os.environ.get('LIDAL_AUTO_ADOPT', '1') != '0'
    _AUTO['on'] = bool(adopt_torch_modules)
    if not _AUTO['on']:
        _AUTO['pending'].clear()
        _disarm()
    return me


# ---- adoption at the first forward call (install_as_torchsparse) ----------------------------------------------------
# Every nn.Conv3d constructed while the switch is on is noted as pending and arms ONE global forward pre-hook; the first
# module called whose tree holds pending convolutions -- the user's model: a root's pre-hook runs before its children's --
# is adopted and the hook removed again (a global hook puts every nn.Module call on torch's slow path, so it must not
# outlive its one job).  Building another model later arms it again.
import weakref as _weakref

_AUTO = {'on': False, 'pending': _weakref.WeakSet(), 'handle': None}


def _note_conv3d(module):
    if _AUTO['on']:
        _AUTO['pending'].add(module)
        if _AUTO['handle'] is None:
            import torch
            _AUTO['handle'] = torch.nn.modules.module.register_module_forward_pre_hook(_adopt_hook)


def _disarm():
    if _AUTO['handle'] is not None:
        _AUTO['handle'].remove()
        _AUTO['handle'] = None


def _adopt_hook(module, args):
    pending = _AUTO['pending']
    if len(pending):
        mine = [m for m in module.modules() if m in pending]
        if mine:
            adopt_torch_modules(module)
            for m in mine:
                pending.discard(m)
    if not len(pending):
        _disarm()
    return None


def adopt_torch_modules(model):
    """Optional second line for a drop-in user (after `install_as_torchsparse()` and building the reference's model):
    hand the model's plain-torch ROW-WISE modules to this package, in place and without touching a single parameter --

        torch.nn.Linear       -> lidal_amd.nn.Linear        (same Parameter objects; the dense kernel, bias in its epilogue,
                                                               weight gradient on the split-K MFMA kernel, bias gradient by column sums)
        torch.nn.BatchNorm1d  -> lidal_amd.nn.BatchNorm1d   (same parameters and buffers; lidal_bn_* kernels)
        nn.Sequential(Linear, BatchNorm1d, ReLU)             -> the same three children with the ReLU inside the BatchNorm kernels
                                                               and the Linear's epilogue leaving the batch statistics

    -- i.e. SPVCNN's point branch and classifier (network/spvcnn.py:60-98), which the reference builds from torch's own
    modules and which cost 7 of the 25 ms of a drop-in training step as rocBLAS / channels-last batch-norm kernels
    (profiles/README.md, round 5).  Module names, Sequential indices and therefore state_dict keys are unchanged;
    exact module TYPES only (a subclass is somebody's own module and is left alone).  Returns the model."""
    import torch
    from . import nn as spnn
    from .network.blocks import ConvNormSequential

    def linear(old):
        new = spnn.Linear(old.in_features, old.out_features, bias=old.bias is not None)
        new.weight = old.weight
        if old.bias is not None:
            new.bias = old.bias
        new.train(old.training)
        return new

    def norm(old):
        new = spnn.BatchNorm1d(old.num_features, eps=old.eps, momentum=old.momentum, affine=old.affine,
                               track_running_stats=old.track_running_stats)
        for k, v in old._parameters.items():
            new._parameters[k] = v
        for k, v in old._buffers.items():
            new._buffers[k] = v
        new.train(old.training)
        return new

    def convert(parent):
        for name, child in list(parent.named_children()):
            if type(child) is torch.nn.Linear:
                setattr(parent, name, linear(child))
            elif type(child) is torch.nn.BatchNorm1d:
                setattr(parent, name, norm(child))
            else:
                convert(child)
            child = getattr(parent, name)
            kids = list(child.children()) if type(child) is torch.nn.Sequential else []
            if (len(kids) == 3 and isinstance(kids[0], spnn.Linear) and type(kids[1]) is spnn.BatchNorm1d
                    and type(kids[2]) is torch.nn.ReLU):
                kids[0].bn_follows = True
                kids[1].fused_relu = True
                seq = ConvNormSequential(kids[0], kids[1], torch.nn.Identity())
                seq.train(child.training)
                setattr(parent, name, seq)
    convert(model)
    return model
