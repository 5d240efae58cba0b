"""Oracle: B-spline KAN linear layer and KAN "convolution" (TEST INFRASTRUCTURE).

Follows convKAN/KANlayers.py:505-660 (KANLinear) and
convKAN/KANConv2Dlayers.py:5-37 (KANConv2d) of the reference: the unfold ->
Cox-de Boor recursion -> two dense products op sequence is kept on purpose,
because this file is also what bench.py times as the CPU baseline ("port").
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

GRID_SIZE = 5
SPLINE_ORDER = 3


def make_grid(in_features, grid_size=GRID_SIZE, spline_order=SPLINE_ORDER, grid_range=(-1.0, 1.0)):
    """Knot buffer [in_features, grid_size + 2*order + 1] (KANlayers.py:526-535)."""
    h = (grid_range[1] - grid_range[0]) / grid_size
    knots = torch.arange(-spline_order, grid_size + spline_order + 1) * h + grid_range[0]
    return knots.expand(in_features, -1).contiguous()


def b_splines(x, grid, spline_order=SPLINE_ORDER):
    """Cox-de Boor bases, half-open knot intervals (KANlayers.py:577-610).

    x [M, in] ; grid [in, G] -> [M, in, G - order - 1].
    """
    x = x.unsqueeze(-1)
    b = ((x >= grid[:, :-1]) & (x < grid[:, 1:])).to(x.dtype)
    for k in range(1, spline_order + 1):
        lo = (x - grid[:, : -(k + 1)]) / (grid[:, k:-1] - grid[:, : -(k + 1)])
        hi = (grid[:, k + 1:] - x) / (grid[:, k + 1:] - grid[:, 1:-k])
        b = lo * b[:, :, :-1] + hi * b[:, :, 1:]
    return b.contiguous()


def kan_linear(x, grid, base_weight, spline_weight, spline_scaler):
    """y = SiLU(x) Wb^T + vec(B(x)) (Ws*scaler)^T  (KANlayers.py:644-660)."""
    out_f = base_weight.shape[0]
    base = F.linear(F.silu(x), base_weight)
    w = spline_weight * spline_scaler.unsqueeze(-1)
    spline = F.linear(b_splines(x, grid).view(x.shape[0], -1), w.view(out_f, -1))
    return base + spline


def kan_conv2d(x, grid, base_weight, spline_weight, spline_scaler, kernel_size=3, stride=1, padding=1):
    """im2col + kan_linear + fold back to NCHW (KANConv2Dlayers.py:15-37).

    Note F.unfold zero-pads *x*, and B-spline bases at x=0 are non-zero, so
    border pixels receive contributions from out-of-image taps.
    """
    b, _, h, w = x.shape
    cols = F.unfold(x, kernel_size=kernel_size, stride=stride, padding=padding)
    cols = cols.transpose(1, 2).reshape(b * cols.shape[2], -1)
    y = kan_linear(cols, grid, base_weight, spline_weight, spline_scaler)
    oh = (h + 2 * padding - kernel_size) // stride + 1
    ow = (w + 2 * padding - kernel_size) // stride + 1
    return y.reshape(b, -1, y.shape[1]).transpose(1, 2).reshape(b, -1, oh, ow)


def regularization_loss(spline_weight, regularize_activation=1.0, regularize_entropy=1.0):
    """KANlayers.py:713-731 (parameter-only surrogate of the KAN paper's L1 + entropy regulariser)."""
    a = spline_weight.abs().mean(dim=-1)
    total = a.sum()
    p = a / total
    return regularize_activation * total - regularize_entropy * (p * p.log()).sum()


class KANLinear(nn.Module):
    """State-dict compatible with reference KANLinear (KANlayers.py:505-575)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("grid", make_grid(in_features))
        self.base_weight = nn.Parameter(torch.empty(out_features, in_features))
        self.spline_weight = nn.Parameter(torch.empty(out_features, in_features, GRID_SIZE + SPLINE_ORDER))
        self.spline_scaler = nn.Parameter(torch.empty(out_features, in_features))
        # Same distributions as KANlayers.py:555-575 except the lstsq-fitted
        # noise spline, replaced by small uniform noise (init is not on the
        # parity path: tests always load explicit weights).
        nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5))
        with torch.no_grad():
            self.spline_weight.uniform_(-0.01, 0.01)

    def forward(self, x):
        return kan_linear(x, self.grid, self.base_weight, self.spline_weight, self.spline_scaler)

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        """KANlayers.py:713-731: a = mean_k |spline_weight| per (out, in) edge; loss = ra * sum(a) + re * H(a / sum(a))."""
        return regularization_loss(self.spline_weight, regularize_activation, regularize_entropy)


class KANConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.kanlayer = KANLinear(in_channels * kernel_size * kernel_size, out_channels)

    def forward(self, x):
        k = self.kanlayer
        return kan_conv2d(x, k.grid, k.base_weight, k.spline_weight, k.spline_scaler,
                          self.kernel_size, self.stride, self.padding)
