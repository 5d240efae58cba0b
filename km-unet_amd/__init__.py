"""km-unet_amd: MI355X-native (gfx950) KM-UNet forward/backward hot path.

The directory name carries a hyphen; import it as ``km_unet_amd`` (root-level km_unet_amd.py
registers this directory under that module name).
"""
import os as _os

# ---- process-wide numerics / runtime policy (only defaults: anything already in the environment wins) ---------------
# MIOpen reads its MIOPEN_DEBUG_* variables when it first needs them and the HIP runtime reads its flags when it
# initialises (first HIP call), so setting them at package import is early enough even after `import torch`
# (measured: tools/env_order_probe.py) -- as long as no convolution has run / no HIP call has been made yet.
#  * MIOPEN_DEBUG_CONV_WINOGRAD=0 -- precaution for fp32 parity: fp32 Winograd transforms carry a larger rounding error
#    than direct / implicit-GEMM convolution and whether MIOpen picks them depends on a timing-based search, i.e. on the
#    box.  None of the glue convolutions in profiles/ ran on a Winograd kernel, so nothing is lost.
#  * DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- the HIP runtime's AQL packet capture for graphs (default on in ROCm 7) replays
#    the multi-thousand-node train-step graph wrongly once the host has synchronised the stream between two replays:
#    reductions inside the captured loss return garbage from then on (measured on MI355X,
#    tools/repro_graph_replay_sync.py: 0.4332 -> 216.4; correct with the capture path off, six other runtime knobs
#    made no difference, profiles/r01c_graph_replay_runtime_knobs.log).  No cost in step time.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import _lib, ops  # noqa: E402,F401
from .model import KM_UNetV3  # noqa: E402,F401
from .nn import DAGEM, DeformConv2d, DySample, EfficientViMBlock, HSMSSD  # noqa: E402,F401
from .nn import IntelligentWaveletPoolingModule, KANConv2d, KANLinear, LayerNorm1D  # noqa: E402,F401

__all__ = ["KM_UNetV3", "KANConv2d", "KANLinear", "HSMSSD", "LayerNorm1D", "EfficientViMBlock", "DySample", "DAGEM",
           "DeformConv2d", "IntelligentWaveletPoolingModule", "ops"]
