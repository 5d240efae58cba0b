"""Graph-replay time of the train step's phases at the bench shape: forward + loss only, and the full step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import train as T
from km_unet_amd.loss import HybridLoss

torch.manual_seed(0)
dev = "cuda"
m = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
data = torch.rand(8, 10, 1, 128, 128, device=dev)
crit = HybridLoss().to(dev)
inp, tgt = T.split_frames(data)
def fwd():
    with torch.no_grad():
        return crit(m(inp), tgt)
def fwd_grad():
    return crit(m(inp), tgt)
def timeit(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, e0.elapsed_time(e1) / 20), flush=True)
timeit(fwd, "forward + loss (no autograd)")
timeit(fwd_grad, "forward + loss (taped)")
