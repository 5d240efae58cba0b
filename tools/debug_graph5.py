"""Eager vs hipGraph-replayed training (with a host sync in the middle): do the PARAMETERS agree?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd import train as T

def two_stage_mean(x):                      # no multi-block (semaphore) reduction: [R, n/R] -> [R] -> scalar
    r = 1024 if x.numel() % 1024 == 0 else 1
    return x.reshape(r, -1).sum(1).sum() / x.numel()

def make(loss_mode):
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).cuda().train()
    for m in model.modules():
        if hasattr(m, "drop_prob"): m.drop_prob = 0.0
    torch.manual_seed(1234)
    data = torch.rand(8, 10, 1, 128, 128, device="cuda")
    st = T.TrainStep(model, data, capturable=True, loss="mse")
    if loss_mode == "two_stage":
        st.criterion = lambda o, t: two_stage_mean((o - t) ** 2)
    return model, data, st

for loss_mode in ("mse", "two_stage"):
    m1, d1, s1 = make(loss_mode)
    le = [s1(d1).item() for _ in range(9)]             # eager: 3 + 6 steps (graph warm-up does 3 eager steps)
    m2, d2, s2 = make(loss_mode)
    gs = T.GraphedTrainStep(s2, d2)
    lg = []
    for i in range(6):
        if i == 2: torch.cuda.current_stream().synchronize()
        lg.append(gs(d2).item())
    worst = max(((p1 - p2).abs().max().item() / (p1.abs().max().item() + 1e-12), k) for (k, p1), (_, p2) in zip(m1.state_dict().items(), m2.state_dict().items()) if p1.is_floating_point())
    print(loss_mode, "eager", " ".join("%.5f" % v for v in le[3:]), "| graph", " ".join("%.5f" % v for v in lg), "| worst param rel diff %.2e %s" % worst)
