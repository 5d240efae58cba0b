import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd.train import TrainStep, split_frames
torch.manual_seed(0)
dev = "cuda"
model = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
torch.manual_seed(1234)
data = torch.rand(8, 10, 1, 128, 128, device=dev)
step = TrainStep(model, data, loss="hybrid")
crit = step.criterion
for i in range(34):
    inp, tgt = split_frames(data)
    step.dp.zero_grad()
    out = model(inp)
    d = out - tgt
    mse = (d * d).mean(); wt = (d * d * torch.exp(2 * tgt)).mean()
    tmin, tmax = torch.aminmax(tgt); pmin, pmax = torch.aminmax(out.detach())
    ss = crit.ssim((out - pmin) / (pmax - pmin + 1e-8), (tgt - tmin) / (tmax - tmin + 1e-8))
    loss = crit(out, tgt)
    loss.backward()
    gn = step.dp.bucket.flat.norm().item()
    step.opt.step()
    if i % 3 == 0 or loss.item() > 5:
        print("step %2d loss %.4f mse %.4f wt %.4f ssim %.4f  pred[min %.3g max %.3g mean %.3g] |grad| %.3g finite=%s" % (
            i, loss.item(), mse.item(), wt.item(), ss.item(), pmin.item(), pmax.item(), out.mean().item(), gn, bool(torch.isfinite(out).all())))
