"""Oracle: Haar wavelet pooling module (TEST INFRASTRUCTURE).

Follows WPL/iwp.py:47-132.  The reference builds banded analysis matrices from
the pywt 'haar' taps (+-1/sqrt 2, mathematical constants) on every call and
applies them as dense mat-muls; that op sequence is kept.  Two quirks are
reproduced on purpose:
* the high-pass matrix loop runs to L1-L-1, so its LAST ROW IS ALL ZERO
  (iwp.py:79) => last row of HL/HH and last column of LH/HH are 0;
* nn.Softmax2d over a ONE-channel map is identically 1 (iwp.py:120-127).
"""
import math

import torch
import torch.nn as nn

_S = 1.0 / math.sqrt(2.0)


def haar_matrices(h, w, dtype, device):
    """-> (low_0 [h/2,h], low_1 [w,w/2], high_0 [h/2,h], high_1 [w,w/2]) (iwp.py:57-104)."""
    assert h % 2 == 0 and w % 2 == 0
    l1 = max(h, w)
    half = l1 // 2
    lo = torch.zeros(half, l1, dtype=torch.float64)
    hi = torch.zeros(l1 - half, l1, dtype=torch.float64)
    for i in range(half):
        lo[i, 2 * i], lo[i, 2 * i + 1] = _S, _S
    for i in range(l1 - half - 1):          # NOTE: one row short, as in the reference
        hi[i, 2 * i], hi[i, 2 * i + 1] = _S, -_S
    cast = lambda m: m.to(dtype=dtype, device=device)
    return (cast(lo[: h // 2 + 1, :h][: h // 2]), cast(lo[: w // 2 + 1, :w][: w // 2].t()),
            cast(hi[: h // 2 + 1, :h][: h // 2]), cast(hi[: w // 2 + 1, :w][: w // 2].t()))


def dwt2d_haar(x):
    """-> LL, LH, HL, HH (DWTFunction_2D.forward, iwp.py:11-26)."""
    h, w = x.shape[-2:]
    l0, l1, h0, h1 = haar_matrices(h, w, x.dtype, x.device)
    lo = torch.matmul(l0, x)
    hi = torch.matmul(h0, x)
    return torch.matmul(lo, l1), torch.matmul(lo, h1), torch.matmul(hi, l1), torch.matmul(hi, h1)


class _DWT(nn.Module):          # parameter-free; kept so module paths match
    def forward(self, x):
        return dwt2d_haar(x)


class IntelligentWaveletPoolingModule(nn.Module):
    def __init__(self, in_channels, wavename="haar"):
        super().__init__()
        assert wavename == "haar"
        self.dwt = _DWT()
        self.high_freq_conv = nn.Conv2d(3 * in_channels, 1, 1)
        self.softmax = nn.Softmax2d()
        self.fusion_conv = nn.Conv2d(in_channels + 1, in_channels, 1)

    def forward(self, x):
        ll, lh, hl, hh = self.dwt(x)
        high = torch.cat([lh, hl, hh], dim=1)
        high = high * self.softmax(self.high_freq_conv(high))        # attention == 1
        return self.fusion_conv(torch.cat([ll, high.mean(dim=1, keepdim=True)], dim=1))
