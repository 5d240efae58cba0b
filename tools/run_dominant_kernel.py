"""Launch the step's dominant hand-written kernels in isolation (for rocprofv3 --pmc passes):
K2 forward + backward at the bench shape (B=8, C=16, 128x128) and K1 forward at site 1."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import ops

torch.manual_seed(0)
d = "cuda"
B, C, Hs, N = 8, 16, 128, 64
x = torch.randn(B, C, Hs * Hs, device=d, requires_grad=True)
w = [torch.randn(3 * N, C, 1, device=d) / 4, torch.randn(3 * N, 1, 3, 3, device=d) * 0.3, torch.randn(2 * C, C, 1, device=d) / 4,
     torch.randn(C, C, 1, device=d) / 4, torch.ones(N, device=d), torch.ones(1, device=d)]
w = [t.requires_grad_(True) for t in w]
gy = torch.randn(B, C, Hs, Hs, device=d)
xk = torch.randn(B, 16, 128, 128, device=d, requires_grad=True)
grid = km_unet_amd.KANLinear(144, 16).grid.to(d)
kw = [t.requires_grad_(True) for t in (torch.randn(16, 144, device=d) * 0.1, torch.randn(16, 144, 8, device=d) * 0.1, torch.randn(16, 144, device=d))]
xc = torch.randn(B, 64, 128, 128, device=d, requires_grad=True)
wc = (torch.randn(16, 64, 3, 3, device=d) * 0.05).requires_grad_(True)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y, h = ops.hsmssd(x, *w)
    y.backward(gy)
    yk = ops.kan_conv2d(xk, grid, *kw)               # K1 forward + input / weight gradients (matrix core)
    yk.backward(gy)
    yc = ops.conv3x3(xc, wc, None)                   # plain 3x3 conv 64 -> 16: forward, dgrad, wgrad
    yc.backward(gy)
torch.cuda.synchronize()
print("done")

# ---- round 4: LayerNorm1D + HSMSSD forward as the two launches of csrc/hsmssd_v2.inc at the three level shapes of the bench
for (Bm, Cm, Hm) in ((8, 16, 128), (8, 32, 64), (24, 64, 32)):
    xm = torch.randn(Bm, Cm, Hm * Hm, device=d, requires_grad=True)
    lw, lb = torch.ones(Cm, device=d, requires_grad=True), torch.zeros(Cm, device=d, requires_grad=True)
    wm = [torch.randn(3 * N, Cm, 1, device=d) / Cm ** 0.5, torch.randn(3 * N, 1, 3, 3, device=d) * 0.3, torch.randn(2 * Cm, Cm, 1, device=d) / Cm ** 0.5,
          torch.randn(Cm, Cm, 1, device=d) / Cm ** 0.5, torch.ones(N, device=d), torch.ones(1, device=d)]
    wm = [t.requires_grad_(True) for t in wm]
    gm = torch.randn(Bm, Cm, Hm, Hm, device=d)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        ym, hm = ops.mixer_ln(xm, lw, lb, 1e-5, *wm)
        ym.backward(gm)          # C = 16: correlation, C-row contractions, gate, pass B on the {B, dt} rows; C >= 32: pass A, gate, pass B
torch.cuda.synchronize()
print("mixer done")

# ---- streaming glue kernels at the bench shapes (for the FETCH_SIZE / WRITE_SIZE passes) --------------------------
import torch.nn as nn
xg = torch.randn(B, 16, 128, 128, device=d, requires_grad=True)
w1 = torch.randn(64, 16, 1, 1, device=d, requires_grad=True)
wdw = torch.randn(16, 1, 3, 3, device=d, requires_grad=True)
bn = nn.BatchNorm2d(16).to(d).train()
alpha = torch.zeros(16, device=d, requires_grad=True)
g64 = torch.randn(B, 64, 128, 128, device=d)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y = ops.pwconv(xg, w1, None)                    # 16 -> 64 @ 128x128: fwd, dgrad, wgrad
    y.backward(g64)
    z = ops.dw_bn_blend(xg, type("C", (), {"weight": wdw})(), bn, alpha)   # dwconv3x3 + BN(train) + blend: fwd / bwd
    z.backward(gy)
# EnhancedViMBlock's tail as one node (TripleNorm + FFN + residual) and the small round-2 glue kernels
C = 16
tn = [torch.randn(C, device=d, requires_grad=True) for _ in range(6)]
w0, b0 = (torch.randn(4 * C, C, 1, 1, device=d) * 0.2).requires_grad_(True), torch.zeros(4 * C, device=d, requires_grad=True)
w2, b2 = (torch.randn(C, 4 * C, 1, 1, device=d) * 0.2).requires_grad_(True), torch.zeros(C, device=d, requires_grad=True)
gate = torch.rand(B, C, device=d, requires_grad=True)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o = ops.VimTailFn.apply(xg, *tn, 1e-5, 1e-5, w0, b0, w2, b2, None)
    o.backward(gy)
    l = ops.lca_apply(xg, gate)
    l.backward(gy)
    m3 = ops.spatial_mean(xg)
torch.cuda.synchronize()
print("glue done")
