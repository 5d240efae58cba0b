"""Oracle: HybridLoss of the reference training loop (TEST INFRASTRUCTURE).

Follows train_shanghai.py:298-325:
    loss = 0.7 * (0.55 * MSE + 0.45 * mean((pred-target)^2 * exp(2*target))) + 0.3 * (1 - SSIM(pred_n, target_n))
with pred_n / target_n min-max normalised by their own (detached) extrema (+1e-8).

SSIM is torchmetrics.image.StructuralSimilarityIndexMeasure(data_range=1.0) (torchmetrics 1.5.2,
requirements.txt:82) -- third-party, not vendored in the reference and not installed here => restated from its
published defaults, PARITY UNPINNED: gaussian 11x11 window, sigma 1.5, k1 0.01, k2 0.03, inputs reflect-padded by
5, per-channel (depthwise) filtering, both variances clamped at 0, the padded border cropped from the SSIM map,
mean over (C,H,W) then batch.
"""
import torch
import torch.nn.functional as F


def gaussian_window(kernel_size=11, sigma=1.5, dtype=torch.float32):
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=dtype)
    g = torch.exp(-((dist / sigma) ** 2) / 2)
    g = g / g.sum()
    return torch.outer(g, g)


def ssim(pred, target, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    c = pred.shape[1]
    pad = (kernel_size - 1) // 2
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    win = gaussian_window(kernel_size, sigma, pred.dtype).to(pred.device).expand(c, 1, kernel_size, kernel_size)
    p = F.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat((p, t, p * p, t * t, p * t))
    out = F.conv2d(stack, win, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(pred.shape[0])
    s_pp, s_tt = (e_pp - mu_p * mu_p).clamp(min=0.0), (e_tt - mu_t * mu_t).clamp(min=0.0)   # variances: non-negative
    s_pt = e_pt - mu_p * mu_t
    smap = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    smap = smap[..., pad:-pad, pad:-pad]
    return smap.reshape(smap.shape[0], -1).mean(-1).mean()


def hybrid_loss(pred, target, alpha=0.7):
    mse = F.mse_loss(pred, target)
    weighted = ((pred - target).pow(2) * torch.exp(target * 2)).mean()
    tn = (target - target.min().detach()) / (target.max().detach() - target.min().detach() + 1e-8)
    pn = (pred - pred.min().detach()) / (pred.max().detach() - pred.min().detach() + 1e-8)
    return alpha * (0.55 * mse + 0.45 * weighted) + (1 - alpha) * (1 - ssim(pn, tn))
