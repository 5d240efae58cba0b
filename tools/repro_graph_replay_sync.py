"""2x2: DropPath on/off x loss kind, graph replay with a stream sync before replay #3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd import train as T
from km_unet_amd.loss import HybridLoss

class Part(HybridLoss):
    def __init__(self, mode):
        super().__init__(); self.mode = mode
    def forward(self, pred, target):
        d = pred - target; sq = d * d
        if self.mode == "mse": return sq.mean()
        if self.mode == "wmse": return (sq * torch.exp(target * 2)).mean()
        if self.mode == "expmean": return sq.mean() + 0 * torch.exp(target * 2).mean()
        if self.mode == "minmax":
            tmin, tmax = torch.aminmax(target.detach()); pmin, pmax = torch.aminmax(pred.detach())
            return sq.mean() + 0 * (tmin + tmax + pmin + pmax) + 1e-3 * (tmax - tmin) + 1e-3 * (pmax - pmin).clamp(max=10.)
        return super().forward(pred, target)

def run(mode, droppath):
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).cuda().train()
    if not droppath:
        for m in model.modules():
            if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device="cuda")
    st = T.TrainStep(model, data, capturable=True, loss="mse")
    st.criterion = Part(mode).cuda()
    gs = T.GraphedTrainStep(st, data)
    vals = []
    for i in range(6):
        if i == 2: torch.cuda.current_stream().synchronize()
        vals.append(gs(data).item())
    print("%-8s droppath=%d " % (mode, droppath), " ".join("%.5f" % v for v in vals), flush=True)

if len(sys.argv) > 1:
    run(sys.argv[1], int(sys.argv[2]))
else:
    for mode in ("mse", "wmse", "minmax", "full"):
        for dp in (0, 1):
            run(mode, dp)
