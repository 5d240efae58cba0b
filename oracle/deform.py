"""Oracle: deformable 3x3 convolution (TEST INFRASTRUCTURE) -- PARITY UNPINNED.

DAGEM_md.py:4,46,101 calls torchvision.ops.DeformConv2d (torchvision 0.14.0,
requirements.txt:83).  torchvision is neither vendored in /root/reference nor
installed in the image, so this restates its published v1 (mask-free)
semantics:

* offset [B, 2*kh*kw, Ho, Wo], channel 2*t = dy and 2*t+1 = dx of tap
  t = i*kw + j (one offset group);
* sample point of tap (i, j) at output (ho, wo):
      y = ho*stride - pad + i*dil + dy ,  x = wo*stride - pad + j*dil + dx ;
* bilinear sample with ZERO outside the image: the whole sample is 0 when
  y <= -1, y >= H, x <= -1 or x >= W, otherwise each of the four corners
  contributes only if it lies inside [0,H-1]x[0,W-1];
* out = weight[Cout, Cin*kh*kw] @ columns + bias.
"""
import math

import torch
import torch.nn as nn


def deform_sample_columns(x, offset, kh=3, kw=3, stride=1, padding=1, dilation=1):
    """-> columns [B, Cin, kh*kw, Ho, Wo] of bilinearly sampled inputs."""
    b, c, h, w = x.shape
    ho, wo = offset.shape[2], offset.shape[3]
    dev, dt = x.device, x.dtype
    off = offset.view(b, kh * kw, 2, ho, wo)
    base_y = (torch.arange(ho, device=dev, dtype=dt) * stride - padding).view(1, 1, ho, 1)
    base_x = (torch.arange(wo, device=dev, dtype=dt) * stride - padding).view(1, 1, 1, wo)
    ti = (torch.arange(kh * kw, device=dev) // kw).to(dt).view(1, kh * kw, 1, 1) * dilation
    tj = (torch.arange(kh * kw, device=dev) % kw).to(dt).view(1, kh * kw, 1, 1) * dilation
    y = base_y + ti + off[:, :, 0]
    xx = base_x + tj + off[:, :, 1]
    inside = (y > -1) & (y < h) & (xx > -1) & (xx < w)
    y0 = torch.floor(y)
    x0 = torch.floor(xx)
    ly, lx = y - y0, xx - x0
    hy, hx = 1 - ly, 1 - lx
    y0 = y0.long()
    x0 = x0.long()
    flat = x.reshape(b, c, h * w)

    def corner(yi, xi, wgt):
        ok = inside & (yi >= 0) & (yi <= h - 1) & (xi >= 0) & (xi <= w - 1)
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).view(b, 1, -1).expand(b, c, -1)
        v = torch.gather(flat, 2, idx).view(b, c, kh * kw, ho, wo)
        return v * (wgt * ok.to(dt)).unsqueeze(1)

    return (corner(y0, x0, hy * hx) + corner(y0, x0 + 1, hy * lx)
            + corner(y0 + 1, x0, ly * hx) + corner(y0 + 1, x0 + 1, ly * lx))


def deform_conv2d(x, offset, weight, bias=None, stride=1, padding=1, dilation=1):
    co, ci, kh, kw = weight.shape
    cols = deform_sample_columns(x, offset, kh, kw, stride, padding, dilation)
    b, _, _, ho, wo = cols.shape
    out = torch.einsum("ok,bkp->bop", weight.reshape(co, ci * kh * kw), cols.reshape(b, ci * kh * kw, ho * wo))
    out = out.view(b, co, ho, wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


class DeformConv2d(nn.Module):
    """Same parameters / init as torchvision.ops.DeformConv2d (weight, bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        assert groups == 1
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_channels * kernel_size * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, offset):
        return deform_conv2d(x, offset, self.weight, self.bias, self.stride, self.padding, self.dilation)
