"""Drop-in for convKAN/KANlayers.py -- only KANLinear (the B-spline layer KM-UNet uses) is provided."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _boot  # noqa: E402,F401
from km_unet_amd.nn import KANLinear  # noqa: E402,F401
