"""K2 (HSMSSD) forward + backward at the three encoder shapes of the bench workload, 10 iterations each -- a target for
`rocprofv3 --kernel-trace --stats` when comparing kernel variants (KMU_LIB_VARIANT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import ops

torch.manual_seed(0)
d, B, N = "cuda", 8, 64
for C, Hs in ((16, 128), (32, 64), (64, 32)):
    x = torch.randn(B, C, Hs * Hs, device=d, requires_grad=True)
    w = [torch.randn(3 * N, C, 1, device=d) / 4, torch.randn(3 * N, 1, 3, 3, device=d) * 0.3, torch.randn(2 * C, C, 1, device=d) / 4,
         torch.randn(C, C, 1, device=d) / 4, torch.ones(N, device=d), torch.ones(1, device=d)]
    w = [t.requires_grad_(True) for t in w]
    gy = torch.randn(B, C, Hs, Hs, device=d)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        y, h = ops.hsmssd(x, *w)
        y.backward(gy)
    torch.cuda.synchronize()
    print("C=%d done, |dx|=%.6e |dW|=%.6e" % (C, x.grad.abs().sum().item(), w[0].grad.abs().sum().item()), flush=True)
