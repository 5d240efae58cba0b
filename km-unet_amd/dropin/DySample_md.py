"""Drop-in for DySample_md.py (KM_UNetV3_SH.py:18)."""
import _boot  # noqa: F401
from km_unet_amd.nn import DySample  # noqa: F401
