#!/usr/bin/env python3
"""Isolated timing of the fused FFN stage (csrc/ffn_fused.hip) at the three bench shapes: per-stage HIP-event times of N forward +
backward calls.  Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
usage: bench_ffn.py [iters] [variant-tag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd  # noqa: E402
from km_unet_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = "cuda"
    torch.manual_seed(0)
    out = {}
    for (B, C, H) in ((8, 16, 128), (8, 32, 64), (8, 64, 32)):
        ffn = km_unet_amd.nn.FFN(C, 4 * C).to(dev).train()
        with torch.no_grad():
            ffn.fc2.norm.weight.fill_(1.0)
        alpha = torch.zeros(C, device=dev, requires_grad=True)
        x = torch.randn(B, C, H, H, device=dev, requires_grad=True)
        gy = torch.randn(B, C, H, H, device=dev)
        for _ in range(3):
            ops.ffn_blend(x, ffn.fc1, ffn.fc2, alpha).backward(gy)
        torch.cuda.synchronize()
        ops.profile_begin()
        for _ in range(iters):
            ops.ffn_blend(x, ffn.fc1, ffn.fc2, alpha).backward(gy)
        prof = ops.profile_end()
        for (name, shape), ms in sorted(prof.items()):
            out["%s%s" % (name, list(shape))] = 1e3 * sum(ms) / len(ms)
    tot = 0.0
    for k, v in out.items():
        print("%-34s %7.1f us" % (k, v))
        tot += v
    print("sum %.1f us" % tot)


if __name__ == "__main__":
    main()
