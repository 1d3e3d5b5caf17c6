"""torchsparse.nn.functional surface used by the reference (network/utils.py:2):
sphash, sphashquery, spcount, spvoxelize, spdevoxelize, calc_ti_weights, spdownsample, conv3d."""
from .conv import KernelMap, build_kernel_map, conv3d
from .count import spcount
from .devoxelize import calc_ti_weights, spdevoxelize, ti_weights_and_index
from .downsample import downsample_pyramid, spdownsample, unique_sorted
from .fused import add_relu, cross_entropy
from .hash import sphash
from .query import HashTable, coords_table, sphashquery
from .voxelize import spvoxelize

__all__ = ['sphash', 'sphashquery', 'HashTable', 'coords_table', 'spcount', 'spvoxelize', 'spdevoxelize',
           'calc_ti_weights', 'ti_weights_and_index', 'spdownsample', 'downsample_pyramid', 'unique_sorted', 'conv3d',
           'add_relu', 'cross_entropy',
           'KernelMap', 'build_kernel_map']
