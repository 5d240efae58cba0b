"""km-unet_amd: MI355X-native (gfx950) KM-UNet forward/backward hot path.

The directory name carries a hyphen; import it as ``km_unet_amd`` (root-level km_unet_amd.py
registers this directory under that module name).
"""
from . import _lib, ops  # noqa: F401
from .model import KM_UNetV3  # noqa: F401
from .nn import (DAGEM, DeformConv2d, DySample, EfficientViMBlock, HSMSSD, IntelligentWaveletPoolingModule,  # noqa: F401
                 KANConv2d, KANLinear, LayerNorm1D)

__all__ = ["KM_UNetV3", "KANConv2d", "KANLinear", "HSMSSD", "LayerNorm1D", "EfficientViMBlock", "DySample", "DAGEM",
           "DeformConv2d", "IntelligentWaveletPoolingModule", "ops"]
