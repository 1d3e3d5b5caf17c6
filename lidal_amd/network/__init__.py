from .unet import SPVCNN, MinkUNet

__all__ = ['SPVCNN', 'MinkUNet']
