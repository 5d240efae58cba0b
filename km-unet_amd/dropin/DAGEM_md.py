"""Drop-in for DAGEM_md.py (KM_UNetV3_SH.py:17)."""
import _boot  # noqa: F401
from km_unet_amd.nn import DAGEM, DeformConv2d  # noqa: F401
