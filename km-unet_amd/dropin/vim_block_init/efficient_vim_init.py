"""Drop-in for vim_block_init/efficient_vim_init.py: HSMSSD and EfficientViMBlock (KM_UNetV3_SH.py:13)."""
from .vim_utils_init import LayerNorm1D  # noqa: F401
from km_unet_amd.nn import EfficientViMBlock, HSMSSD  # noqa: F401
