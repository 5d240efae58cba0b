"""BatchNorm-blend forward + backward at the bench shapes, 10 iterations each (a rocprofv3 --stats target for kernel variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import km_unet_amd
from km_unet_amd import ops
d = "cuda"
for C, hw in ((16, 128), (64, 128), (32, 64), (128, 64), (64, 32)):
    t = torch.randn(8, C, hw, hw, device=d, requires_grad=True); x = torch.randn(8, C, hw, hw, device=d, requires_grad=True)
    bn = nn.BatchNorm2d(C).to(d).train(); alpha = torch.zeros(C, device=d, requires_grad=True)
    g = torch.randn(8, C, hw, hw, device=d)
    for _ in range(10):
        y = ops.bn_blend(t, x, bn, alpha)
        torch.autograd.grad(y, (t, x), g)
    torch.cuda.synchronize()
    print("C=%d %d done %.6e" % (C, hw, y.abs().sum().item()), flush=True)
