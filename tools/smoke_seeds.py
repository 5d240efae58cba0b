"""smoke()'s comparison for several random inputs in ONE process / environment: is the 1.1e-3 input-gradient error a
property of the kernels (every seed) or of one input (a discontinuity tie)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from oracle.model import KM_UNetV3 as Oracle, fill_parameters
o = fill_parameters(Oracle(num_classes=5), 2).eval()
m = km_unet_amd.KM_UNetV3(num_classes=5)
m.load_state_dict(o.state_dict(), strict=True)
m = m.to("cuda:0").eval()
for seed in range(8):
    torch.manual_seed(seed)
    x = torch.rand(2, 5, 32, 32); tgt = torch.rand(2, 5, 32, 32)
    xo = x.clone().requires_grad_(True)
    o.zero_grad(); torch.nn.functional.mse_loss(o(xo), tgt).backward()
    xg = x.to("cuda:0").requires_grad_(True)
    m.zero_grad(); torch.nn.functional.mse_loss(m(xg), tgt.to("cuda:0")).backward()
    d = xg.grad.cpu() - xo.grad
    print("seed %d  dx max %.2e  L2 %.2e" % (seed, (d.abs().max() / xo.grad.abs().max()).item(), (d.norm() / xo.grad.norm()).item()))
