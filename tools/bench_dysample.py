#!/usr/bin/env python3
"""HIP-event times of DySample forward / backward at the three decoder levels (B = 8, C = 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import ops
from oracle import dysample as od
d = "cuda"
for H in (16, 32, 64):
    x = torch.randn(8, 64, H, H, device=d, requires_grad=True)
    conv = (torch.randn(8, 32, H, H, device=d) * 0.01).requires_grad_(True)
    gy = torch.randn(8, 64, 2 * H, 2 * H, device=d)
    ip = od.init_pos().to(d)
    for _ in range(3):
        ops.dysample_lp(x, conv, ip).backward(gy)
    ops.profile_begin()
    for _ in range(10):
        ops.dysample_lp(x, conv, ip).backward(gy)
    pr = ops.profile_end()
    print(H, {k[0]: round(1e3 * sum(v) / len(v), 1) for k, v in pr.items()})
