"""Per-shape timing of the BatchNorm-blend kernels (back-to-back launches): forward (stats + apply) and backward (reduce + apply)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import km_unet_amd
from km_unet_amd import ops
d = "cuda"
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, hw in ((16, 128), (64, 128), (32, 64), (128, 64), (64, 32)):
    t = torch.randn(8, C, hw, hw, device=d); x = torch.randn(8, C, hw, hw, device=d)
    bn = nn.BatchNorm2d(C).to(d).train(); alpha = torch.zeros(C, device=d, requires_grad=True)
    tr = t.clone().requires_grad_(True); xr = x.clone().requires_grad_(True)
    with torch.no_grad():
        tf = timeit(lambda: ops.bn_blend(t, x, bn, alpha))
    y = ops.bn_blend(tr, xr, bn, alpha); g = torch.randn_like(y)
    tb = timeit(lambda: torch.autograd.grad(y, (tr, xr), g, retain_graph=True))
    print("C=%3d %3dx%-3d  fwd (2 kernels) %6.1f us   bwd (2 kernels + autograd) %6.1f us" % (C, hw, hw, tf, tb), flush=True)
