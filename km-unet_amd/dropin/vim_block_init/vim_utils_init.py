"""Drop-in for vim_block_init/vim_utils_init.py (the pieces EfficientViMBlock needs)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _boot  # noqa: E402,F401
from km_unet_amd.nn import ConvLayer1D, ConvLayer2D, FFN, LayerNorm1D  # noqa: E402,F401
