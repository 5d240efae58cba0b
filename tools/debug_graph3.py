import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import torch.nn.functional as F
import km_unet_amd
from km_unet_amd import train as T
from km_unet_amd.loss import HybridLoss

class Part(HybridLoss):
    def __init__(self, mode):
        super().__init__(); self.mode = mode
    def forward(self, pred, target):
        d = pred - target; sq = d * d
        if self.mode == "wmse":
            return 0.7 * (0.55 * sq.mean() + 0.45 * (sq * torch.exp(target * 2)).mean())
        if self.mode == "minmax":
            tmin, tmax = torch.aminmax(target.detach()); pmin, pmax = torch.aminmax(pred.detach())
            return (((target - tmin) / (tmax - tmin + 1e-8)) - ((pred - pmin) / (pmax - pmin + 1e-8))).pow(2).mean()
        if self.mode == "ssim":
            return 0.3 * (1 - self.ssim(pred, target))
        return super().forward(pred, target)

def run(tag, mode, sync):
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).cuda().train()
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device="cuda")
    eager = T.TrainStep(model, data, capturable=True, loss="mse")
    eager.criterion = Part(mode).cuda()
    gs = T.GraphedTrainStep(eager, data)
    vals = []
    for i in range(6):
        if i == 2:
            if sync == "device": torch.cuda.synchronize()
            if sync == "stream": torch.cuda.current_stream().synchronize()
        vals.append(gs(data).item())
    print("%-28s" % tag, " ".join("%.4f" % v for v in vals))

w = sys.argv[1]
if w == "1": run("wmse + device sync", "wmse", "device")
if w == "2": run("minmax + device sync", "minmax", "device")
if w == "3": run("ssim + device sync", "ssim", "device")
if w == "4": run("hybrid + stream sync", "full", "stream")
