"""Accuracy of every conv configuration the model still sends to MIOpen (F.conv2d, fp32) against fp64 on the CPU:
forward, input gradient, weight gradient.  Run under different MIOPEN_DEBUG_* settings (set before python starts)."""
import os, sys
import torch
import torch.nn.functional as F
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # model input size (32 = smoke, 128 = bench)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfgs = [("conv_f", 5, 16, 3, S), ("dec2.1", 64, 32, 3, S // 2), ("dec3.1", 64, 16, 3, S), ("dec3.3", 16, 5, 3, S),
        ("msf1.b0", 16, 32, 3, S // 4), ("msf1.b1", 32, 32, 5, S // 4), ("msf1.b2", 32, 32, 7, S // 4), ("msf1.f1", 32, 32, 3, S // 4),
        ("msf2.b0", 16, 32, 3, S // 2), ("msf2.b1", 32, 32, 5, S // 2), ("msf2.b2", 32, 32, 7, S // 2), ("msf2.f1", 32, 32, 3, S // 2),
        ("dagem.off", 64, 18, 3, S // 8)]
print("env:", {k: v for k, v in os.environ.items() if k.startswith("MIOPEN_")})
for name, ci, co, k, hw in cfgs:
    x = torch.randn(B, ci, hw, hw, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(co, ci, k, k, dtype=torch.float64) / (ci * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(co, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(B, co, hw, hw, dtype=torch.float64)
    y = F.conv2d(x, w, b, padding=k // 2); y.backward(gy)
    xd, wd, bd = [t.detach().float().cuda().requires_grad_(True) for t in (x, w, b)]
    yd = F.conv2d(xd, wd, bd, padding=k // 2); yd.backward(gy.float().cuda())
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()
    e = (rel(yd.detach(), y.detach()), rel(xd.grad, x.grad), rel(wd.grad, w.grad))
    print("%-10s %3d->%3d k%d %3dx%-3d  y %.1e  dx %.1e  dw %.1e %s" % (name, ci, co, k, hw, hw, *e, "  <== " if max(e) > 1e-4 else ""))
