"""Drop-in for convKAN/KANConv2Dlayers.py -- KANConv2d; the 8 alternative KAN variants the reference
comments out (KM_UNetV3_SH.py:28-32) are out of scope."""
from .KANlayers import KANLinear  # noqa: F401
from km_unet_amd.nn import KANConv2d  # noqa: F401
