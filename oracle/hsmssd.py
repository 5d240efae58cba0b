"""Oracle: EfficientViM hidden-state-mixer block (TEST INFRASTRUCTURE).

Follows vim_block_init/efficient_vim_init.py:14-97 (HSMSSD, EfficientViMBlock)
and vim_block_init/vim_utils_init.py:34-130 (LayerNorm1D, ConvLayer1D/2D, FFN).
Module attribute names reproduce the reference state_dict keys.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def layernorm1d(x, weight, bias, eps=1e-5):
    """Per-token LN over the channel axis of [B,C,L], biased variance
    (vim_utils_init.py:50-59)."""
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def hsmssd(x, w_bcdt, w_dw, w_hz, w_out, A, D, state_dim):
    """HSM-SSD mixer (efficient_vim_init.py:33-61).

    x [B,C,L] (L a perfect square) ; w_bcdt [3N,C,1] ; w_dw [3N,1,3,3] ;
    w_hz [2C,C,1] ; w_out [C,C,1] ; A [N] ; D [1]  ->  y [B,C,H,H], h [B,C,N].
    There is no recurrence: softmax over the token axis + two batched products.
    """
    b, c, L = x.shape
    hh = int(math.sqrt(L))
    n = state_dim
    p = F.conv1d(x, w_bcdt)                                     # :39  1x1 projection
    bcdt = F.conv2d(p.view(b, 3 * n, hh, hh), w_dw, padding=1, groups=3 * n).flatten(2)
    Bm, Cm, dt = torch.split(bcdt, [n, n, n], dim=1)            # :41
    a = (dt + A.view(1, -1, 1)).softmax(-1)                     # :46  (shift-invariant => A is a no-op)
    h = x @ (a * Bm).transpose(-2, -1)                          # :48-50  [B,C,N]
    hz = F.conv1d(h, w_hz)                                      # :52
    h1, z = torch.split(hz, [c, c], dim=1)
    h2 = F.conv1d(h1 * F.silu(z) + h1 * D, w_out)               # :55
    y = h2 @ Cm                                                 # :57
    return y.view(b, c, hh, hh).contiguous(), h2


class LayerNorm1D(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(1, c, 1))
        self.bias = nn.Parameter(torch.zeros(1, c, 1))

    def forward(self, x):
        return layernorm1d(x, self.weight, self.bias, self.eps)


class _Conv1D(nn.Module):
    """ConvLayer1D with norm=None, act=None (the only form HSMSSD uses)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, 1, bias=False)

    def forward(self, x):
        return self.conv(x)


class ConvBN2D(nn.Module):
    """ConvLayer2D (vim_utils_init.py:62-89): bias-free conv [+ BatchNorm2d] [+ ReLU]."""

    def __init__(self, cin, cout, k=3, padding=0, groups=1, norm=True, act=True, bn_weight_init=1.0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, 1, padding, groups=groups, bias=False)
        self.norm = nn.BatchNorm2d(cout) if norm else None
        self.act = nn.ReLU() if act else None
        if norm:
            nn.init.constant_(self.norm.weight, bn_weight_init)
            nn.init.constant_(self.norm.bias, 0)

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.act is not None:
            x = self.act(x)
        return x


class HSMSSD(nn.Module):
    def __init__(self, d_model, ssd_expand=1, A_init_range=(1, 16), state_dim=64):
        super().__init__()
        assert ssd_expand == 1
        self.state_dim = state_dim
        self.BCdt_proj = _Conv1D(d_model, 3 * state_dim)
        self.dw = ConvBN2D(3 * state_dim, 3 * state_dim, 3, 1, groups=3 * state_dim, norm=False, act=False)
        self.hz_proj = _Conv1D(d_model, 2 * d_model)
        self.out_proj = _Conv1D(d_model, d_model)
        self.A = nn.Parameter(torch.empty(state_dim).uniform_(*A_init_range))
        self.D = nn.Parameter(torch.ones(1))

    def forward(self, x):
        return hsmssd(x, self.BCdt_proj.conv.weight, self.dw.conv.weight, self.hz_proj.conv.weight,
                      self.out_proj.conv.weight, self.A, self.D, self.state_dim)


class FFN(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.fc1 = ConvBN2D(in_dim, dim, 1)
        self.fc2 = ConvBN2D(dim, in_dim, 1, act=False, bn_weight_init=0.0)

    def forward(self, x):
        return self.fc2(self.fc1(x))


class EfficientViMBlock(nn.Module):
    """efficient_vim_init.py:64-97: sigmoid-alpha blends around dw3x3+BN, the
    mixer, dw3x3+BN and the 1x1 FFN."""

    def __init__(self, dim, mlp_ratio=4.0, ssd_expand=1, state_dim=64):
        super().__init__()
        self.mixer = HSMSSD(dim, ssd_expand, state_dim=state_dim)
        self.norm = LayerNorm1D(dim)
        self.dwconv1 = ConvBN2D(dim, dim, 3, 1, groups=dim, act=False, bn_weight_init=0.0)
        self.dwconv2 = ConvBN2D(dim, dim, 3, 1, groups=dim, act=False, bn_weight_init=0.0)
        self.ffn = FFN(dim, int(dim * mlp_ratio))
        self.alpha = nn.Parameter(1e-4 * torch.ones(4, dim))

    def forward(self, x):
        a = torch.sigmoid(self.alpha).view(4, -1, 1, 1)
        x = (1 - a[0]) * x + a[0] * self.dwconv1(x)
        y, _ = self.mixer(self.norm(x.flatten(2)))
        x = (1 - a[1]) * x + a[1] * y
        x = (1 - a[2]) * x + a[2] * self.dwconv2(x)
        x = (1 - a[3]) * x + a[3] * self.ffn(x)
        return x
