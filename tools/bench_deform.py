#!/usr/bin/env python3
"""Isolated timing of DAGEM's deformable conv (sampling + contraction) at the bridge shape [8,64,16,16]; run under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations.  usage: bench_deform.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd  # noqa: E402,F401
from km_unet_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
B, C, H = 8, 64, 16
x = torch.randn(B, C, H, H, device="cuda", requires_grad=True)
off = (0.5 * torch.randn(B, 18, H, H, device="cuda")).requires_grad_(True)
w = (0.1 * torch.randn(C, C, 3, 3, device="cuda")).requires_grad_(True)
gy = torch.randn(B, C, H, H, device="cuda")
for _ in range(3):
    ops.deform_conv2d(x, off, w).backward(gy)
torch.cuda.synchronize()
ops.profile_begin()
for _ in range(iters):
    ops.deform_conv2d(x, off, w).backward(gy)
prof = ops.profile_end()
for (name, shape), ms in sorted(prof.items()):
    print("%-40s %7.1f us" % ("%s%s" % (name, list(shape)), 1e3 * sum(ms) / len(ms)))
