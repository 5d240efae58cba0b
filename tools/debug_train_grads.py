"""Debug probe (GPU): localise the train-mode gradient discrepancy of the whole model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from conftest import load_golden, rel_err
from oracle.model import KM_UNetV3 as Oracle, fill_parameters

g = load_golden("model_sh_train")
o = fill_parameters(Oracle(num_classes=5), 1).double().train()
for sub in o.modules():
    if hasattr(sub, "drop_prob"): sub.drop_prob = 0.0
x64 = g["x"].double().requires_grad_(True)
torch.nn.functional.mse_loss(o(x64), g["target"].double()).backward()
ref = {k: p.grad for k, p in o.named_parameters() if p.grad is not None}

def run():
    m = fill_parameters(km_unet_amd.KM_UNetV3(num_classes=5), 1).cuda().train()
    for sub in m.modules():
        if hasattr(sub, "drop_prob"): sub.drop_prob = 0.0
    x = g["x"].cuda().requires_grad_(True)
    torch.nn.functional.mse_loss(m(x), g["target"].cuda()).backward()
    return x.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}

runs = [run() for _ in range(3)]
for i, (dx, gr) in enumerate(runs):
    errs = sorted(((rel_err(gr[k], ref[k]), k) for k in gr if not k.endswith(".A")), reverse=True)
    print("run", i, "dx err %.2e" % rel_err(dx, x64.grad), " worst:", [(("%.1e" % e), k[-60:]) for e, k in errs[:6]])
d01 = max(((runs[0][1][k] - runs[1][1][k]).abs().max().item() / (runs[0][1][k].abs().max().item() + 1e-30), k) for k in runs[0][1])
print("run-to-run max rel diff:", d01, " dx:", ((runs[0][0] - runs[1][0]).abs().max() / runs[0][0].abs().max()).item())
# per-module error by prefix (mean of per-tensor errors)
import collections
agg = collections.defaultdict(list)
for k in runs[0][1]:
    if k.endswith(".A"): continue
    agg[".".join(k.split(".")[:2])].append(rel_err(runs[0][1][k], ref[k]))
print({k: "%.1e" % max(v) for k, v in agg.items()})
