"""Drop-in for the reference's KM_UNetV3_SH.py: `from KM_UNetV3_SH import KM_UNetV3` (train_shanghai.py:7)."""
import _boot  # noqa: F401
from km_unet_amd.model import (ChannelAttention, DirectionAttention, DirectionViM, EnhancedViMBlock,  # noqa: F401
                               LocalContrastAttention, MultiScaleFusion, StableHybridKANConv, TripleNorm)
from km_unet_amd.model import KM_UNetV3 as _Base


class KM_UNetV3(_Base):
    def __init__(self, num_classes=3, embed_dims=[16, 32, 64]):
        super().__init__(num_classes=num_classes, embed_dims=embed_dims, variant="SH")
