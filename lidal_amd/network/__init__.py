from .geometry import Geometry, GeometryPrefetcher
from .unet import SPVCNN, MinkUNet

__all__ = ['SPVCNN', 'MinkUNet', 'Geometry', 'GeometryPrefetcher']
