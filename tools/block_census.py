"""Kernel list (torch.profiler, device time) of one sub-module's forward + backward at its bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
dev = "cuda"
m = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
name = sys.argv[1]
mod, shape = {"msf1": (None, None), "bridge": (m.bridge_attention, (8, 64, 16, 16)), "lca1": (m.lca1, (8, 16, 64, 64)), "kan1": (m.enc1[0], (8, 16, 128, 128)),
              "iwp1": (m.enc1[2], (8, 16, 128, 128)), "dec1": (m.dec1, (8, 64, 16, 16)), "vim32": (m.enc2[1], (8, 32, 64, 64)), "vim64": (m.enc3[1], (8, 64, 32, 32)),
              "vim16": (m.enc1[1], (8, 16, 128, 128)), "kan3": (m.enc3[0], (8, 32, 32, 32))}[name]
if name == "msf1":
    feats = [torch.randn(8, c, 32, 32, device=dev, requires_grad=True) for c in (16, 32, 32)]
    inner = m.attention1[0]
    params = [p for p in inner.parameters() if p.requires_grad]
    def step():
        y = inner(list(feats))
        return torch.autograd.grad(y.float().square().mean(), feats + params, allow_unused=True)
else:
    x = torch.randn(*shape, device=dev, requires_grad=True)
    params = [p for p in mod.parameters() if p.requires_grad]
    def step():
        y = mod(x)
        return torch.autograd.grad(y.float().square().mean(), [x] + params, allow_unused=True)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt > 0 and e.device_type.name != "CPU":
        rows.append((dt, e.count, e.key[:110]))
rows.sort(reverse=True)
print("%s: %d kernels, %.1f us device time" % (name, sum(r[1] for r in rows), sum(r[0] for r in rows)))
for dt, n, k in rows[:int(os.environ.get("TOP", "60"))]:
    print("%8.1f us %3d x  %s" % (dt, n, k))
