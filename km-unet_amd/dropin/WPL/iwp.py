"""Drop-in for WPL/iwp.py (KM_UNetV3_SH.py:16)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _boot  # noqa: E402,F401
from km_unet_amd.nn import IntelligentWaveletPoolingModule  # noqa: E402,F401
