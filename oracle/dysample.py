"""Oracle: DySample x2 dynamic upsampler, style 'lp' (TEST INFRASTRUCTURE).

Follows DySample_md.py:45-68.  Two forms are given:

* ``dysample_lp``          -- the reference op sequence ending in
                              F.grid_sample(bilinear, border, align_corners=False);
* ``dysample_lp_indices``  -- explicit index generation + gather, restating the
                              fp32 op order of ATen's CPU grid_sample
                              (unnormalise ((g+1)*W-1)/2, clip to [0,W-1],
                              floor, corner weights).  This is the integer
                              oracle the HIP kernel must match bit-exactly.
"""
import torch
import torch.nn.functional as F


def init_pos(scale=2, groups=4):
    """DySample_md.py:45-47 -> [1, 2*groups*scale^2, 1, 1]."""
    h = torch.arange((-scale + 1) / 2, (scale - 1) / 2 + 1) / scale
    return torch.stack(torch.meshgrid([h, h], indexing="ij")).transpose(1, 2).repeat(1, groups, 1).reshape(1, -1, 1, 1)


def dysample_offset(x, w_off, b_off, ipos):
    """forward_lp without scope (DySample_md.py:67): conv1x1 * 0.25 + init_pos."""
    return F.conv2d(x, w_off, b_off) * 0.25 + ipos


def normalized_coords(offset, scale=2):
    """DySample_md.py:50-59 -> grid [B*g, sH, sW, 2] in grid_sample convention."""
    b, _, h, w = offset.shape
    off = offset.view(b, 2, -1, h, w)
    ch = torch.arange(h, dtype=offset.dtype, device=offset.device) + 0.5
    cw = torch.arange(w, dtype=offset.dtype, device=offset.device) + 0.5
    coords = torch.stack(torch.meshgrid([cw, ch], indexing="ij")).transpose(1, 2).unsqueeze(1).unsqueeze(0)
    norm = torch.tensor([w, h], dtype=offset.dtype, device=offset.device).view(1, 2, 1, 1, 1)
    coords = 2 * (coords + off) / norm - 1
    coords = F.pixel_shuffle(coords.reshape(b, -1, h, w), scale).view(b, 2, -1, scale * h, scale * w)
    return coords.permute(0, 2, 3, 4, 1).contiguous().flatten(0, 1)


def dysample_lp(x, w_off, b_off, ipos, scale=2, groups=4):
    b, c, h, w = x.shape
    grid = normalized_coords(dysample_offset(x, w_off, b_off, ipos), scale)
    out = F.grid_sample(x.reshape(b * groups, -1, h, w), grid, mode="bilinear",
                        align_corners=False, padding_mode="border")
    return out.view(b, -1, scale * h, scale * w)


def sample_indices(grid, h, w):
    """grid [..., 2] normalised -> (ix0, iy0 int32 ; fx, fy fp32 fractional parts).

    Un-normalise (align_corners=False): p = ((g + 1) * size - 1) / 2 ;
    border padding: clip p to [0, size-1]; i0 = floor(p).  Corner 1 (= i0+1)
    is used only when it is <= size-1 (its weight is exactly 0 otherwise).
    """
    gx, gy = grid[..., 0], grid[..., 1]
    px = ((gx + 1) * w - 1) / 2
    py = ((gy + 1) * h - 1) / 2
    px = px.clamp(0, w - 1)
    py = py.clamp(0, h - 1)
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    return x0.to(torch.int32), y0.to(torch.int32), px - x0, py - y0


def dysample_lp_indices(x, w_off, b_off, ipos, scale=2, groups=4):
    """Index-explicit restatement; returns (out, ix0, iy0) with the index
    tensors shaped [B*g, sH, sW]."""
    b, c, h, w = x.shape
    grid = normalized_coords(dysample_offset(x, w_off, b_off, ipos), scale)
    ix0, iy0, fx, fy = sample_indices(grid, h, w)
    xg = x.reshape(b * groups, c // groups, h * w)
    x0 = ix0.long()
    y0 = iy0.long()
    x1 = (x0 + 1).clamp(max=w - 1)
    y1 = (y0 + 1).clamp(max=h - 1)

    def g(yy, xx):
        idx = (yy * w + xx).view(b * groups, 1, -1).expand(-1, c // groups, -1)
        return torch.gather(xg, 2, idx).view(b * groups, c // groups, scale * h, scale * w)

    fx = fx.unsqueeze(1)
    fy = fy.unsqueeze(1)
    out = (g(y0, x0) * (1 - fx) * (1 - fy) + g(y0, x1) * fx * (1 - fy)
           + g(y1, x0) * (1 - fx) * fy + g(y1, x1) * fx * fy)
    return out.view(b, c, scale * h, scale * w), ix0, iy0


class DySample(torch.nn.Module):
    def __init__(self, in_channels, scale=2, style="lp", groups=4, dyscope=False):
        super().__init__()
        assert style == "lp" and not dyscope
        self.scale, self.groups = scale, groups
        self.offset = torch.nn.Conv2d(in_channels, 2 * groups * scale ** 2, 1)
        torch.nn.init.normal_(self.offset.weight, 0, 0.001)
        torch.nn.init.constant_(self.offset.bias, 0)
        self.register_buffer("init_pos", init_pos(scale, groups))

    def forward(self, x):
        return dysample_lp(x, self.offset.weight, self.offset.bias, self.init_pos, self.scale, self.groups)
