"""Drop-in for convKAN/KANConv2Dlayers.py.

`KANConv2d` (KANConv2Dlayers.py:5-37) is the layer KM-UNet uses and runs on the HIP kernels.  The eight alternative
basis families of the file (:40-293 -- Chebyshev, FastKAN, GRAM, wavelet, Jacobi, ReLU-KAN, FasterKAN, RBF) are
commented out in the reference's model (KM_UNetV3_SH.py:28-32) and are not on the accelerated path: `from
convKAN.KANConv2Dlayers import *` still resolves every name, and constructing one says so instead of failing with an
ImportError somewhere else (same constructor signatures as the reference)."""
import torch.nn as nn

from .KANlayers import KANLinear  # noqa: F401
from km_unet_amd.nn import KANConv2d  # noqa: F401


def _variant(name, lines, extra=""):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, **kw):
        raise NotImplementedError(
            "%s (convKAN/KANConv2Dlayers.py:%s) is not part of the MI355X hot path: KM_UNetV3 uses KANConv2d only "
            "(the alternatives are commented out at KM_UNetV3_SH.py:28-32).  Use KANConv2d, or the reference's own "
            "PyTorch class for experiments with this basis.%s" % (name, lines, extra))
    return type(name, (nn.Module,), {"__init__": __init__, "__doc__": "Not accelerated; see module docstring."})


ChebyKANConv2d = _variant("ChebyKANConv2d", "40-68", " (ctor: ..., degree=4)")
FastKANConv2d = _variant("FastKANConv2d", "71-100")
GRAMKANConv2d = _variant("GRAMKANConv2d", "103-133")
WavKANConv2d = _variant("WavKANConv2d", "136-165", " (ctor: ..., wavelet_type='mexican_hat')")
JacobiKANConv2d = _variant("JacobiKANConv2d", "168-198", " (ctor: ..., degree=4)")
ReLUKANConv2d = _variant("ReLUKANConv2d", "201-231")
FasterKANConv2d = _variant("FasterKANConv2d", "234-263")
RBFKANConv2d = _variant("RBFKANConv2d", "266-293")

__all__ = ["KANConv2d", "KANLinear", "ChebyKANConv2d", "FastKANConv2d", "GRAMKANConv2d", "WavKANConv2d", "JacobiKANConv2d",
           "ReLUKANConv2d", "FasterKANConv2d", "RBFKANConv2d"]
