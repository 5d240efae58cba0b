"""smoke()'s comparison with more metrics; run under different KMU_GLUE_TORCH settings to bisect a gradient discrepancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from oracle.model import KM_UNetV3 as Oracle, fill_parameters
torch.manual_seed(0)
o = fill_parameters(Oracle(num_classes=5), 2).eval()
m = km_unet_amd.KM_UNetV3(num_classes=5)
m.load_state_dict(o.state_dict(), strict=True)
m = m.to("cuda:0").eval()
x = torch.rand(2, 5, 32, 32); tgt = torch.rand(2, 5, 32, 32)
xo = x.clone().requires_grad_(True)
torch.nn.functional.mse_loss(o(xo), tgt).backward()
xg = x.to("cuda:0").requires_grad_(True)
torch.nn.functional.mse_loss(m(xg), tgt.to("cuda:0")).backward()
d = (xg.grad.cpu() - xo.grad); ref = xo.grad
mx = ref.abs().max()
print("glue=%-22s dx: max %.2e  l2 %.2e  frac>1e-4 %.2e  argmax %s" % (os.environ.get("KMU_GLUE_TORCH", "-"), (d.abs().max() / mx).item(),
      (d.norm() / ref.norm()).item(), (d.abs() > 1e-4 * mx).float().mean().item(), tuple(int(i) for i in torch.nonzero(d.abs() == d.abs().max())[0])))
po = dict(o.named_parameters()); worst = []
for k, p in m.named_parameters():
    if p.grad is not None and po[k].grad is not None:
        g, r = p.grad.cpu(), po[k].grad
        worst.append(((g - r).abs().max().item() / (r.abs().max().item() + 1e-30), k))
worst = [(e, k) for e, k in worst if not k.endswith("mixer.A")]      # d/dA is exactly 0 here, ~1e-7 noise in the oracle
worst.sort(reverse=True)
print("   worst param grads:", ["%s %.1e" % (k, e) for e, k in worst[:8]])
print("   params with err > 2e-4: %d of %d" % (sum(1 for e, _ in worst if e > 2e-4), len(worst)))
