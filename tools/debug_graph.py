import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd.loss import HybridLoss
from km_unet_amd.train import TrainStep, GraphedTrainStep
dev = "cuda"
# (2) whole step, replay-by-replay
for loss_kind in ("hybrid",):
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device=dev)
    eager = TrainStep(model, data, capturable=True, loss=loss_kind)
    gs = GraphedTrainStep(eager, data)
    vals = []
    for i in range(45):
        vals.append(gs(data).item())
    with torch.no_grad():
        out = model.eval()(data.squeeze(2)[:, :5])
        print('eval pred min/max/mean', out.min().item(), out.max().item(), out.mean().item(), 'finite', bool(torch.isfinite(out).all()))
        bad = [k for k, v in model.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
        print('non-finite state entries:', bad[:8])
    print(loss_kind, "graph replays:", " ".join("%.4f" % v for v in vals))
