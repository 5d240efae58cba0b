"""Drop-in for convKAN/KANConv2Dlayers.py.

`KANConv2d` (KANConv2Dlayers.py:5-37) is the layer KM-UNet uses and runs on the HIP kernels.  The eight alternative basis families of
the file (:40-293 -- Chebyshev, FastKAN, GRAM, wavelet, Jacobi, ReLU-KAN, FasterKAN, RBF) are commented out in the reference's model
(KM_UNetV3_SH.py:28-32) and are not on the accelerated path: they are pass-through PyTorch modules with the reference's constructor
signatures, state_dict keys and numerics (km-unet_amd/kan_variants.py; pinned by tests/golden/kanvar_*.npz)."""
from .KANlayers import KANLinear  # noqa: F401
from km_unet_amd.nn import KANConv2d  # noqa: F401
from km_unet_amd.kan_variants import (ChebyKANConv2d, FastKANConv2d, FasterKANConv2d, GRAMKANConv2d, JacobiKANConv2d,  # noqa: F401
                                      RBFKANConv2d, ReLUKANConv2d, WavKANConv2d)

__all__ = ["KANConv2d", "KANLinear", "ChebyKANConv2d", "FastKANConv2d", "GRAMKANConv2d", "WavKANConv2d", "JacobiKANConv2d",
           "ReLUKANConv2d", "FasterKANConv2d", "RBFKANConv2d"]
