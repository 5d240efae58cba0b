"""Training-step harness reproducing the reference loop's contract (train_shanghai.py:163-181):
data [B,T,1,H,W] -> squeeze(2) -> input = data[:, :5], target = data[:, 5:] -> model -> loss ->
backward -> optimizer step.  AdamW(lr 1e-3, weight_decay 0.05) as train_shanghai.py:342.

Loss: the reference's HybridLoss (train_shanghai.py:298-325; km-unet_amd/loss.py) by default, plain MSE on request.
"""
import math

import torch
import torch.nn.functional as F

from . import ops
from .dp import DataParallel, live_parameters
from .loss import HybridLoss


def split_frames(data):
    data = data.squeeze(2)
    # packed copies: MIOpen serves strided (non-packed) conv inputs with its naive kernels
    return data[:, :5].float().contiguous(), data[:, 5:].float().contiguous()


class TrainStep:
    def __init__(self, model, example_data, lr=1e-3, weight_decay=0.05, process_group=None, capturable=False,
                 loss="hybrid", force_collective=False):
        self.model = model
        self.criterion = HybridLoss().to(example_data.device) if loss == "hybrid" else F.mse_loss
        inp, _ = split_frames(example_data)
        live = live_parameters(model, inp)
        self.dp = DataParallel(model, live, process_group, force_collective)
        kw = dict(lr=lr, weight_decay=weight_decay)
        if example_data.is_cuda:
            kw.update(fused=True, capturable=capturable)
            if capturable:      # the learning rate lives in a device tensor: a schedule can change it under a captured graph
                kw["lr"] = torch.tensor(float(lr), device=example_data.device, dtype=torch.float32)
        # one flat parameter tensor (its .grad is the flat gradient bucket): AdamW is a single launch
        self.flat_param = self.dp.bucket.flatten_parameters()
        self.opt = torch.optim.AdamW([self.flat_param], **kw)

    def forward_backward(self, data):
        inp, tgt = split_frames(data)
        # the weights do not change between here and the optimizer: every split-bf16 weight pack is made once (ops.PackCache)
        with ops.pack_scope():
            ops.prepack()
            out = self.model(inp)
            loss = self.criterion(out, tgt)
            self.dp.backward(loss)
        return loss

    def __call__(self, data):
        loss = self.forward_backward(data)
        self.dp.all_reduce_grads()
        self.opt.step()
        return loss


class CosineAnnealing:
    """torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=200, eta_min=5e-4), stepped once per epoch
    (train_shanghai.py:398-399, :415), in closed form:  eta_min + (base - eta_min) (1 + cos(pi t / T_max)) / 2.
    When the optimizer holds its learning rate in a device tensor (TrainStep(capturable=True)) the new value is written
    IN PLACE, so a captured hipGraph picks it up at its next replay; a float learning rate is simply replaced."""

    def __init__(self, optimizer, T_max=200, eta_min=5e-4):
        self.opt, self.T_max, self.eta_min, self.last_epoch = optimizer, T_max, eta_min, 0
        self.base = [float(g["lr"]) for g in optimizer.param_groups]

    def lr_at(self, epoch):
        return [self.eta_min + (b - self.eta_min) * (1.0 + math.cos(math.pi * epoch / self.T_max)) / 2.0 for b in self.base]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.opt.param_groups, self.lr_at(self.last_epoch)):
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def get_last_lr(self):
        return [float(g["lr"]) for g in self.opt.param_groups]


class GraphedTrainStep:
    """The same step replayed from captured hipGraphs (torch.cuda.CUDAGraph): ~4.7k kernel launches per step
    otherwise make the loop host-bound.  One process: a single graph holds zero-grad, forward, loss, backward
    and the fused AdamW.  Several processes: graph 1 = zero-grad + forward + backward, then the (eager) RCCL
    all-reduce of the flat gradient bucket, then graph 2 = AdamW -- the collective stays outside the capture.
    Requires TrainStep(..., capturable=True); shapes are static; DropPath draws from the captured Philox stream.

    Self-check (`validate=True`, the default): before it is handed out, the captured step is replayed three times with
    a host synchronisation between replays and each loss is compared with ONE eager step from the same weights,
    optimizer state and buffers (restored afterwards, so construction leaves the training state where it found it).
    Round 1 met a runtime whose graph replays returned garbage from the third replay on (loss 0.43 -> 216, see
    DESIGN.md section 5); a graph that does not reproduce the eager step raises here instead of training on noise.
    `check_loss()` repeats the cheap part (finite, no jump against the previous call) at any later point."""

    def __init__(self, step, example_data, warmup=3, validate=True):
        # The captured step forks ~9 streams (3 direction branches, 2 pyramids, 4 weight-gradient lanes).  With GPU_MAX_HW_QUEUES
        # below the runtime's default of 4 the HIP runtime (ROCm 7.2) dies with SIGSEGV inside hipGraphLaunch at the FIRST replay
        # (python -X faulthandler: torch/cuda/graphs.py replay <- _validate; no kernel of ours involved; seen twice with the knob
        # at 2, rounds 2 and 3) -- a drop-in library inherits its host's environment, so refuse with a message instead.
        import os
        q = os.environ.get("GPU_MAX_HW_QUEUES")
        if q is not None and q.strip().isdigit() and int(q) < 4:
            raise RuntimeError("GraphedTrainStep: GPU_MAX_HW_QUEUES=%s -- the HIP runtime crashes (SIGSEGV in hipGraphLaunch) when a graph with "
                               "this step's parallel branches is replayed on fewer than 4 hardware queues; unset the variable (the default "
                               "is 4, measured fastest) or train with the eager TrainStep" % q)
        self.step = step
        self.static_data = example_data.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step(self.static_data)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        snap = self._snapshot() if validate else None
        self.g1 = torch.cuda.CUDAGraph()
        self.g2 = None
        # capture_error_mode="thread_local": with a process group up, RCCL's watchdog thread keeps polling its work events
        # (hipEventQuery) -- under the default "global" mode such a call from ANOTHER thread during the capture invalidates it
        # and aborts the process (seen with the 1-rank RCCL test, intermittently: it depends on whether the warm-up steps'
        # all-reduce work has been retired yet)
        mode = dict(capture_error_mode="thread_local")
        if not step.dp.collective:
            with torch.cuda.graph(self.g1, **mode):
                self.loss = step(self.static_data)
        else:
            with torch.cuda.graph(self.g1, **mode):
                self.loss = step.forward_backward(self.static_data)
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2, **mode):
                step.opt.step()
        self._last = None
        if validate:
            self._validate(snap)

    # ---- state that one step changes: parameters (flat), AdamW moments + step counter, module buffers (BatchNorm) ----
    def _state_tensors(self):
        ts = [self.step.flat_param.data, self.step.dp.bucket.flat]
        for st in self.step.opt.state.values():
            ts += [v for v in st.values() if torch.is_tensor(v)]
        ts += [b for b in self.step.model.buffers()]
        return ts

    def _snapshot(self):
        return [t.detach().clone() for t in self._state_tensors()], torch.cuda.get_rng_state()

    def _restore(self, snap):
        with torch.no_grad():
            for t, s in zip(self._state_tensors(), snap[0]):
                t.copy_(s)
        torch.cuda.set_rng_state(snap[1])

    def _validate(self, snap):
        self._restore(snap)
        ref = float(self.step(self.static_data))                 # eager step from the snapshot
        # DropPath masks of a replay need not equal the eager step's (different Philox offsets): its per-sample
        # Bernoulli(0.9) scaling moves the loss by a few per cent at most, garbage moves it by orders of magnitude
        tol = 0.25 * abs(ref) + 1e-3
        self._restore(snap)
        seen = []
        for i in range(3):
            torch.cuda.current_stream().synchronize()
            seen.append(float(self(self.static_data)))
            if i == 0 and not (abs(seen[0] - ref) <= tol):
                break
        ok = all(v == v and abs(v - ref) <= tol for v in seen)
        self._restore(snap)
        self._last = None
        if not ok:
            raise RuntimeError("GraphedTrainStep: the captured step does not reproduce the eager step (eager loss %.6g, replays %s). "
                               "Run eagerly (TrainStep) or see DESIGN.md section 5 (DEBUG_CLR_GRAPH_PACKET_CAPTURE)." % (ref, seen))

    def check_loss(self, factor=4.0):
        """Host-side sanity check of the last replay's loss (one device sync): finite, and not `factor` times the loss at
        the previous check.  Cheap enough for every logging interval of a training loop."""
        v = float(self.loss)
        if not (v == v) or v in (float("inf"), float("-inf")) or (self._last is not None and v > factor * self._last + 1e-3):
            raise RuntimeError("GraphedTrainStep: loss %.6g after %.6g -- the replayed graph no longer computes the step" % (v, self._last or float("nan")))
        self._last = v
        return v

    def __call__(self, data):
        if data is not self.static_data:
            self.static_data.copy_(data)
        self.g1.replay()
        if self.g2 is not None:
            self.step.dp.all_reduce_grads()
            self.g2.replay()
        return self.loss
