"""Drop-in for convKAN/KANlayers.py: KANLinear (the B-spline layer KM-UNet uses, HIP kernels behind KANConv2d) and the row-wise layers
of the eight alternative KAN convolutions (pass-through PyTorch modules, km-unet_amd/kan_variants.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _boot  # noqa: E402,F401
from km_unet_amd.nn import KANLinear  # noqa: E402,F401
from km_unet_amd.kan_variants import (ChebyKANLayer, FastKANLayer, FasterKANLayer, GRAMLayer, JacobiKANLayer, RBFKANLayer, RBFLinear,  # noqa: E402,F401
                                      ReLUKANLayer, SplineLinear, SplineLinear_fstr, WavKANLayer)
