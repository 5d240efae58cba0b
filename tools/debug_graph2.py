import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd.train import TrainStep, GraphedTrainStep

def run(tag, set_device, dev, sync_between, copy_data=True):
    if set_device:
        torch.cuda.set_device(0)
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device=dev)
    eager = TrainStep(model, data, capturable=True, loss="hybrid")
    gs = GraphedTrainStep(eager, data)
    vals = []
    for i in range(6):
        if i == 2 and sync_between:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        vals.append(gs(data if copy_data else gs.static_data).item())
    print(tag, " ".join("%.4f" % v for v in vals))

which = sys.argv[1]
if which == "a": run("set_device+dev(cuda,0)+sync", True, torch.device("cuda", 0), True)
if which == "b": run("no set_device, 'cuda', sync", False, "cuda", True)
if which == "c": run("set_device+dev(cuda,0), no sync", True, torch.device("cuda", 0), False)
if which == "d": run("set_device, no data copy", True, torch.device("cuda", 0), True, copy_data=False)
