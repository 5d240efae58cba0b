"""Training-step harness reproducing the reference loop's contract (train_shanghai.py:163-181):
data [B,T,1,H,W] -> squeeze(2) -> input = data[:, :5], target = data[:, 5:] -> model -> loss ->
backward -> optimizer step.  AdamW(lr 1e-3, weight_decay 0.05) as train_shanghai.py:342.

Loss: plain MSE for now; the reference's HybridLoss (train_shanghai.py:298-325) adds a weighted-MSE and
a torchmetrics SSIM term (third-party, restated in a later round -- SURVEY 8f-1).
"""
import torch
import torch.nn.functional as F

from .dp import DataParallel, live_parameters


def split_frames(data):
    data = data.squeeze(2)
    return data[:, :5].float(), data[:, 5:].float()


class TrainStep:
    def __init__(self, model, example_data, lr=1e-3, weight_decay=0.05, process_group=None, capturable=False):
        self.model = model
        inp, _ = split_frames(example_data)
        live = live_parameters(model, inp)
        self.dp = DataParallel(model, live, process_group)
        kw = dict(lr=lr, weight_decay=weight_decay)
        if example_data.is_cuda:
            kw.update(fused=True, capturable=capturable)
        self.opt = torch.optim.AdamW(live, **kw)

    def __call__(self, data):
        inp, tgt = split_frames(data)
        self.dp.zero_grad()
        out = self.model(inp)
        loss = F.mse_loss(out, tgt)
        loss.backward()
        self.dp.all_reduce_grads()
        self.opt.step()
        return loss
