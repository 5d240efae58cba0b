"""HybridLoss of the reference training loop (train_shanghai.py:298-325) and the CSI / POD / FAR / HSS
contingency scores of its evaluator (metrics.py:45-47,105-114,220-288) on the device (callers either side of the hot path,
SURVEY.md 8f).  On the GPU HybridLoss runs as the fused kernels of csrc/hybrid_loss.hip + csrc/gauss11.hip; the tensor-op
formulation below is the same arithmetic and serves CPU tensors, other window sizes and targets that need a gradient.

SSIM: torchmetrics' StructuralSimilarityIndexMeasure(data_range=1.0) is third-party and not available here; its
published defaults are restated (gaussian 11x11, sigma 1.5, k1 .01, k2 .03, reflect pad 5, border cropped).  The
11x11 window is applied separably (two 1-D depthwise passes), which is the same filter.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class HybridLoss(nn.Module):
    def __init__(self, alpha=0.7, kernel_size=11, sigma=1.5):
        super().__init__()
        self.alpha, self.k, self.pad = alpha, kernel_size, (kernel_size - 1) // 2
        dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1)
        g = torch.exp(-((dist / sigma) ** 2) / 2)
        self.register_buffer("gauss", g / g.sum(), persistent=False)
        self._taps = {}

    def _filter(self, x):
        if x.is_cuda and self.k == 11:           # HIP separable window (csrc/gauss11.hip); MIOpen has only naive kernels here
            from . import ops
            return ops.gauss11(x, self.gauss)
        c = x.shape[1]
        key = (c, x.device)
        if key not in self._taps:      # packed per-channel copies (a stride-0 expand() sends MIOpen to its naive kernels)
            g = self.gauss.to(x.device)
            self._taps[key] = (g.view(1, 1, self.k, 1).repeat(c, 1, 1, 1), g.view(1, 1, 1, self.k).repeat(c, 1, 1, 1))
        gh, gw = self._taps[key]
        return F.conv2d(F.conv2d(x, gh, groups=c), gw, groups=c)

    def ssim(self, p, t):
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        pad = self.pad
        p = F.pad(p, (pad, pad, pad, pad), mode="reflect")
        t = F.pad(t, (pad, pad, pad, pad), mode="reflect")
        b = p.shape[0]
        mu_p, mu_t, e_pp, e_tt, e_pt = self._filter(torch.cat((p, t, p * p, t * t, p * t))).split(b)
        s_pp, s_tt = (e_pp - mu_p * mu_p).clamp(min=0.0), (e_tt - mu_t * mu_t).clamp(min=0.0)   # variances: non-negative
        s_pt = e_pt - mu_p * mu_t
        smap = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
        smap = smap[..., pad:-pad, pad:-pad]
        return smap.reshape(b, -1).mean(-1).mean()

    def forward(self, pred, target):
        if pred.is_cuda and self.k == 11 and min(pred.shape[-2:]) >= 12 and not target.requires_grad:
            from . import ops
            return ops.hybrid_loss(pred, target, self.gauss, self.alpha)      # fused: 6 launches forward, 3 backward
        _refuse_unsafe_capture(pred)
        d = pred - target
        sq = d * d
        mse = sq.mean()
        weighted = (sq * torch.exp(target * 2)).mean()
        tmin, tmax = torch.aminmax(target.detach())
        pmin, pmax = torch.aminmax(pred.detach())
        tn = (target - tmin) / (tmax - tmin + 1e-8)
        pn = (pred - pmin) / (pmax - pmin + 1e-8)
        return self.alpha * (0.55 * mse + 0.45 * weighted) + (1 - self.alpha) * (1 - self.ssim(pn, tn))


def _refuse_unsafe_capture(t):
    """The tensor-op formulation above, captured into a hipGraph together with DropPath's bernoulli_, is the ONE program
    whose replays went wrong on ROCm 7.2 with the runtime's AQL packet capture enabled (third replay onwards: loss 0.43 ->
    216; tools/graph_replay_bisect.py --loss aten).  The fused kernels (ops.hybrid_loss) replay
    correctly with or without it.  Capturing this path with packet capture on is therefore refused instead of risked."""
    import os
    from . import PACKET_CAPTURE_MAY_BE_ON
    on = PACKET_CAPTURE_MAY_BE_ON or os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1") != "0"
    if on and t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("HybridLoss: the tensor-op fallback must not be captured into a hipGraph while the HIP runtime's graph "
                           "packet capture is enabled (set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the first HIP call, or use the "
                           "fused path: CUDA tensors, 11-tap window, images >= 12x12, target without requires_grad)")


THRESHOLDS = (20, 30, 35, 40)


def contingency_counts(pred, target, thresholds=THRESHOLDS, scale=90):
    """-> int64 tensor [len(thresholds), 4] = (TP, FN, FP, TN) pooled over every element, on the inputs' device.
    CUDA tensors: ONE HIP reduction (csrc/contingency.hip) and no host synchronisation; the uint16 truncation of
    clip(x,0,1)*scale is the reference's (metrics.py:45-47).  CPU tensors: the same counts with tensor ops."""
    n = pred.numel()
    if pred.is_cuda:
        import ctypes
        from . import _lib, ops
        p = ops._f32c(pred.detach(), "pred").reshape(-1)
        t = ops._f32c(target.detach(), "target").reshape(-1)
        k = len(thresholds)
        counts = torch.zeros(k, 3, device=p.device, dtype=torch.int64)
        th = (ctypes.c_int * k)(*[int(v) for v in thresholds])
        _lib.check(_lib.load().kmu_contingency_counts(p.data_ptr(), t.data_ptr(), counts.data_ptr(), n, th, k, float(scale),
                                                      ops._stream()), "kmu_contingency_counts")
    else:
        pi = (pred.detach().float().clamp(0, 1) * scale).to(torch.int32).reshape(1, -1)
        ti = (target.detach().float().clamp(0, 1) * scale).to(torch.int32).reshape(1, -1)
        th = torch.tensor(list(thresholds), dtype=torch.int32).view(-1, 1)
        pb, tb = pi >= th, ti >= th
        counts = torch.stack(((pb & tb).sum(1), (~pb & tb).sum(1), (pb & ~tb).sum(1)), dim=1)
    return torch.cat((counts, n - counts.sum(1, keepdim=True)), dim=1)


def scores_from_counts(counts, thresholds=THRESHOLDS):
    """{threshold: {csi, pod, far, hss}} from a [T,4] (TP, FN, FP, TN) count table (metrics.py:258-264; the reference divides
    unguarded, a zero denominator gives nan).  One device-to-host copy when `counts` lives on the GPU."""
    out = {}
    div = lambda a, b: float(a) / b if b else float("nan")
    for th, (tp, fn, fp, tn) in zip(thresholds, counts.tolist()):
        out[th] = {"csi": div(tp, tp + fp + fn), "pod": div(tp, tp + fn), "far": div(fp, tp + fp),
                   "hss": div(2 * (tp * tn - fp * fn), fp ** 2 + fn ** 2 + 2 * tp * tn + (fp + fn) * (tp + tn))}
    return out


def contingency_scores(pred, target, thresholds=THRESHOLDS, scale=90):
    """CSI / POD / FAR / HSS per threshold: one reduction on the device + a single D2H copy of the 4x4 count table.
    An evaluator over many batches sums contingency_counts() tables on the device and calls scores_from_counts() once."""
    return scores_from_counts(contingency_counts(pred, target, thresholds, scale), thresholds)
