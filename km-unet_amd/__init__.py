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
#  * DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- precaution, not a dependency.  Round 1 saw hipGraph replays of the train step
#    return garbage from the third replay on (loss 0.43 -> 216) with the runtime's AQL packet capture (default on in
#    ROCm 7).  Bisected in round 2 (tools/graph_replay_bisect.py, profiles/r02_graph_replay_bisect.log): the step built
#    from this library's kernels, including the fused HybridLoss, replays correctly WITH packet capture on in every
#    configuration tried; the failure needs the tensor-op HybridLoss fallback (ATen reductions / reflect pad / cat) plus
#    DropPath's captured bernoulli_ in the same graph.  That fallback now refuses to be captured with packet capture on
#    (loss.py), and GraphedTrainStep validates every captured graph against an eager step before handing it out
#    (train.py).  The default below only matters for callers that import this package before the first HIP call.
import torch as _torch

# True when the HIP runtime was already up (so it has read its flags) without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: the default
# set below then comes too late, and loss.py's capture guard must treat packet capture as ON.
PACKET_CAPTURE_MAY_BE_ON = _torch.cuda.is_initialized() and _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0"
_os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")
if not PACKET_CAPTURE_MAY_BE_ON:
    _os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import _lib, ops  # noqa: E402,F401
from .model import KM_UNetV3  # noqa: E402,F401
from .nn import DAGEM, DeformConv2d, DySample, EfficientViMBlock, HSMSSD  # noqa: E402,F401
from .nn import IntelligentWaveletPoolingModule, KANConv2d, KANLinear, LayerNorm1D  # noqa: E402,F401

__all__ = ["KM_UNetV3", "KANConv2d", "KANLinear", "HSMSSD", "LayerNorm1D", "EfficientViMBlock", "DySample", "DAGEM",
           "DeformConv2d", "IntelligentWaveletPoolingModule", "ops"]
