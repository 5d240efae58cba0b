import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd.loss import HybridLoss
from km_unet_amd.train import TrainStep, GraphedTrainStep
dev = "cuda"
# (1) HybridLoss alone: eager vs captured
torch.manual_seed(0)
crit = HybridLoss().to(dev)
pred = torch.rand(8, 5, 128, 128, device=dev, requires_grad=True); tgt = torch.rand(8, 5, 128, 128, device=dev)
le = crit(pred, tgt); le.backward(); ge = pred.grad.clone(); pred.grad = None
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        l = crit(pred, tgt); l.backward(); pred.grad = None
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
pred.grad = torch.zeros_like(pred)
with torch.cuda.graph(g):
    lg = crit(pred, tgt); lg.backward()
pred.grad.zero_(); g.replay(); torch.cuda.synchronize()
print("HybridLoss alone: eager %.6f graph %.6f  grad diff %.2e" % (le.item(), lg.item(), (pred.grad - ge).abs().max().item()))
# (2) whole step, replay-by-replay
for loss_kind in ("hybrid", "mse"):
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device=dev)
    eager = TrainStep(model, data, capturable=True, loss=loss_kind)
    gs = GraphedTrainStep(eager, data)
    vals = []
    for i in range(12):
        vals.append(gs(data).item())
    print(loss_kind, "graph replays:", " ".join("%.4f" % v for v in vals))
