"""Feasibility timing for an un-fused K2 pass B: the generic split-bf16 conv kernels at pass B's shapes
(dx = conv^T(dQ[192 rows]) and dW' = dQ (*) x) next to the fused exact-fp32 hsm_bwd_passB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import ops

d = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 8
for C, Hs in ((16, 128), (32, 64), (64, 32)):
    dq = torch.randn(B, 192, Hs, Hs, device=d)
    w = (torch.randn(C, 192, 3, 3, device=d) * 0.05).requires_grad_(True)     # "forward" conv 192 -> C == dgrad of the composite conv
    x = torch.randn(B, C, Hs, Hs, device=d, requires_grad=True)
    wc = (torch.randn(192, C, 3, 3, device=d) * 0.05).requires_grad_(True)
    with torch.no_grad():
        t_fwd = timeit(lambda: ops.conv3x3(dq, w, None))
    y = ops.conv3x3(x, wc, None)
    def wg():
        torch.autograd.grad(y, wc, dq, retain_graph=True)
    t_wg = timeit(wg)
    def dg():
        torch.autograd.grad(y, x, dq, retain_graph=True)
    t_dg = timeit(dg)
    print("C=%d Hs=%d: conv 192->C (pack+fwd) %.1f us | dgrad path of C->192 (pack+fwd) %.1f us | wgrad C->192 %.1f us" % (C, Hs, t_fwd, t_dg, t_wg), flush=True)
