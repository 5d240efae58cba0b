"""Training-step harness reproducing the reference loop's contract (train_shanghai.py:163-181):
data [B,T,1,H,W] -> squeeze(2) -> input = data[:, :5], target = data[:, 5:] -> model -> loss ->
backward -> optimizer step.  AdamW(lr 1e-3, weight_decay 0.05) as train_shanghai.py:342.

Loss: the reference's HybridLoss (train_shanghai.py:298-325; km-unet_amd/loss.py) by default, plain MSE on request.
"""
import torch
import torch.nn.functional as F

from .dp import DataParallel, live_parameters
from .loss import HybridLoss


def split_frames(data):
    data = data.squeeze(2)
    # packed copies: MIOpen serves strided (non-packed) conv inputs with its naive kernels
    return data[:, :5].float().contiguous(), data[:, 5:].float().contiguous()


class TrainStep:
    def __init__(self, model, example_data, lr=1e-3, weight_decay=0.05, process_group=None, capturable=False,
                 loss="hybrid"):
        self.model = model
        self.criterion = HybridLoss().to(example_data.device) if loss == "hybrid" else F.mse_loss
        inp, _ = split_frames(example_data)
        live = live_parameters(model, inp)
        self.dp = DataParallel(model, live, process_group)
        kw = dict(lr=lr, weight_decay=weight_decay)
        if example_data.is_cuda:
            kw.update(fused=True, capturable=capturable)
        # one flat parameter tensor (its .grad is the flat gradient bucket): AdamW is a single launch
        self.flat_param = self.dp.bucket.flatten_parameters()
        self.opt = torch.optim.AdamW([self.flat_param], **kw)

    def forward_backward(self, data):
        inp, tgt = split_frames(data)
        out = self.model(inp)
        loss = self.criterion(out, tgt)
        self.dp.backward(loss)
        return loss

    def __call__(self, data):
        loss = self.forward_backward(data)
        self.dp.all_reduce_grads()
        self.opt.step()
        return loss


class GraphedTrainStep:
    """The same step replayed from captured hipGraphs (torch.cuda.CUDAGraph): ~4.7k kernel launches per step
    otherwise make the loop host-bound.  One process: a single graph holds zero-grad, forward, loss, backward
    and the fused AdamW.  Several processes: graph 1 = zero-grad + forward + backward, then the (eager) RCCL
    all-reduce of the flat gradient bucket, then graph 2 = AdamW -- the collective stays outside the capture.
    Requires TrainStep(..., capturable=True); shapes are static; DropPath draws from the captured Philox stream."""

    def __init__(self, step, example_data, warmup=3):
        self.step = step
        self.static_data = example_data.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step(self.static_data)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g1 = torch.cuda.CUDAGraph()
        self.g2 = None
        if step.dp.world == 1:
            with torch.cuda.graph(self.g1):
                self.loss = step(self.static_data)
        else:
            with torch.cuda.graph(self.g1):
                self.loss = step.forward_backward(self.static_data)
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2):
                step.opt.step()

    def __call__(self, data):
        if data is not self.static_data:
            self.static_data.copy_(data)
        self.g1.replay()
        if self.g2 is not None:
            self.step.dp.all_reduce_grads()
            self.g2.replay()
        return self.loss
