import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, km_unet_amd
from km_unet_amd import ops
torch.manual_seed(0)
for (B, C, Hs) in ((1, 16, 32), (1, 16, 64)):
    N = 64
    x = torch.randn(B, C, Hs * Hs, device="cuda", requires_grad=True)
    w = [(torch.randn(3 * N, C, 1, device="cuda") / C ** 0.5).requires_grad_(True), (torch.randn(3 * N, 1, 3, 3, device="cuda") * 0.5).requires_grad_(True),
         (torch.randn(2 * C, C, 1, device="cuda") / C ** 0.5).requires_grad_(True), (torch.randn(C, C, 1, device="cuda") / C ** 0.5).requires_grad_(True),
         torch.zeros(N, device="cuda"), torch.ones(1, device="cuda", requires_grad=True)]
    gy = torch.randn(B, C, Hs, Hs, device="cuda")
    out = {}
    for mode in ("f32", "bf16x3"):
        ops.K2_MATH = mode
        y, h = ops.hsmssd(x, *w)
        (dx,) = torch.autograd.grad((y * gy).sum(), [x])
        out[mode] = dx.view(B, C, Hs, Hs)
    e = (out["bf16x3"] - out["f32"]).abs() / out["f32"].abs().max()
    print("case", (B, C, Hs), "max rel err", e.max().item())
    bad = e > 1e-3
    print(" fraction bad", bad.float().mean().item())
    print(" bad per channel:", bad.float().mean(dim=(0, 2, 3)).cpu().numpy().round(2))
    print(" bad per row y  :", bad.float().mean(dim=(0, 1, 3)).cpu().numpy().round(2))
    print(" bad per col x  :", bad.float().mean(dim=(0, 1, 2)).cpu().numpy().round(2))
    r = (out["bf16x3"] / out["f32"])[0, :4, :4, :8]
    print(" ratio sample:", r.cpu().numpy().round(3))
