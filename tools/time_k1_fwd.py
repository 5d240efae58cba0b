#!/usr/bin/env python3
"""Device time of the K1 forward (KANConv2d, csrc/conv3x3_x3.hip) per launch at the four live sites of KM_UNetV3_SH, B = 8, back to back
between one pair of HIP events, for each Cout-split mode of the small-image dispatch (kmu_conv_debug_split: 1 never, 0 automatic, 2 always)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd  # noqa: E402
from km_unet_amd import _lib, ops  # noqa: E402
from km_unet_amd.nn import KANConv2d  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
for mode in (1, 0, 2):
    lib.kmu_conv_debug_split(mode)
    row = {}
    for (cin, cout, hw) in ((16, 16, 128), (16, 32, 64), (32, 64, 32), (64, 32, 32)):
        m = KANConv2d(cin, cout, 3, 1, 1).cuda()
        x = torch.randn(8, cin, hw, hw, device="cuda")
        with torch.no_grad(), ops.pack_scope():
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(30):
                m(x)
            e.record()
            torch.cuda.synchronize()
        row["%d->%d@%d" % (cin, cout, hw)] = round(s.elapsed_time(e) / 30 * 1e3, 1)
    print("split mode", mode, row, flush=True)
lib.kmu_conv_debug_split(0)
