"""nn.Module surface of the hot blocks -- same constructor arguments, attribute names and
state_dict keys as the reference modules they stand in for; forward() launches the HIP kernels
(km-unet_amd/ops.py).  Glue that the reference leaves to stock ATen (convs, BatchNorm, ...)
stays stock PyTorch-ROCm here as well.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


import os as _os

# Debug only (tests/tools): comma list of fused glue ops to route back through stock PyTorch, e.g.
# KMU_GLUE_TORCH=bn_blend,dwconv,conv1x1,qkv_gate -- used to bisect numerics; never set in production.
_TORCH_GLUE = set(filter(None, _os.environ.get("KMU_GLUE_TORCH", "").split(",")))
# DAGEM around its deformable convolution as one launch per BatchNorm boundary (csrc/dagem_fused.hip); False: the round-2 sequence of
# pointwise-conv / BatchNorm / edge kernels (kept as the A/B partner of tests/test_gpu_kernels.py::test_dagem_fused_matches_unfused)
_DAGEM_FUSED = True


def conv1x1(x, conv, gelu_in=False):
    """1x1 convolution of an nn.Conv2d's parameters, optionally of GELU(x): the HIP pointwise-conv kernels
    (csrc/pwconv.hip: bias / GELU / GELU' / bias-gradient folded in) when the channel counts are multiples of 16,
    else ONE strided-batched GEMM  W[Co,Ci] @ x[b][Ci, H*W]  (MIOpen serves NCHW 1x1 convs with im2col + batched
    transposes around a GEMM: 3-4 launches)."""
    if "conv1x1" in _TORCH_GLUE:
        return conv(F.gelu(x) if gelu_in else x)
    b, ci, h, w = x.shape
    co = conv.weight.shape[0]
    if "pwconv" not in _TORCH_GLUE and ops.pwconv_supported(ci, co, h * w):
        return ops.pwconv(x, conv.weight, conv.bias, gelu_in)
    if gelu_in:
        x = F.gelu(x)
    y = torch.bmm(conv.weight.view(1, co, ci).expand(b, co, ci), x.reshape(b, ci, h * w))
    if conv.bias is not None:
        y = y + conv.bias.view(1, co, 1)
    return y.view(b, co, h, w)


def conv3x3(x, conv):
    """A plain nn.Conv2d: dense K x K / stride 1 / padding K//2 with K in {3, 5, 7} -- every such conv of the model, incl.
    MultiScaleFusion's 5x5 / 7x7 -- runs forward and backward on the split-bf16 matrix-core kernels (csrc/conv3x3_x3.hip;
    MIOpen served them with NCHW<->NHWC transposes around an implicit GEMM chosen by a timing search).  KMU_GLUE_TORCH=conv3x3
    keeps MIOpen."""
    k = conv.kernel_size[0]
    if ("conv3x3" not in _TORCH_GLUE and x.is_cuda and conv.kernel_size in ((3, 3), (5, 5), (7, 7)) and conv.stride == (1, 1)
            and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and conv.groups == 1):
        return ops.conv_kxk(x, conv.weight, conv.bias)
    return conv(x)


def gate_mlp(p, lin1, lin2, act1, act2="sigmoid"):
    """act2(lin2(act1(lin1(p)))) for pooled [B, I] vectors; lin1/lin2 are nn.Linear or 1x1 nn.Conv2d modules
    (csrc/gate_mlp.hip: one launch each way).  KMU_GLUE_TORCH=gate_mlp keeps the ATen sequence."""
    if "gate_mlp" in _TORCH_GLUE:
        f = {"gelu": F.gelu, "silu": F.silu, "relu": F.relu}[act1]
        z = F.linear(f(F.linear(p, lin1.weight.flatten(1), lin1.bias)), lin2.weight.flatten(1), lin2.bias)
        return torch.sigmoid(z) if act2 == "sigmoid" else torch.softmax(z, dim=1)
    return ops.gate_mlp(p, lin1.weight, lin1.bias, lin2.weight, lin2.bias, act1, act2)


def group_norm(x, gn, silu=False, sigmoid=False):
    """nn.GroupNorm (+ SiLU: MultiScaleFusion's blocks; + sigmoid: the output head) through the HIP kernels (csrc/group_norm.hip);
    KMU_GLUE_TORCH=group_norm keeps ATen's."""
    if "group_norm" in _TORCH_GLUE or not x.is_cuda:
        y = gn(x)
        return F.silu(y) if silu else (torch.sigmoid(y) if sigmoid else y)
    return ops.group_norm(x, gn, silu, sigmoid)


def _is_pointwise(c):
    return c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0) and c.groups == 1


# ------------------------------------------------------------------ convKAN (K1)
class KANLinear(nn.Module):
    """convKAN/KANlayers.py:505-731.  KM-UNet uses it through KANConv2d, which runs the layer on the HIP kernels (csrc/conv3x3_x3.hip /
    kan_conv2d.hip); the row-wise methods below (b_splines, forward, curve2coeff, update_grid) are stock tensor arithmetic for API
    completeness."""

    def __init__(self, in_features, out_features, grid_size=5, spline_order=3, scale_noise=0.1, scale_base=1.0,
                 scale_spline=1.0, enable_standalone_scale_spline=True, base_activation=nn.SiLU, grid_eps=0.02,
                 grid_range=(-1, 1)):
        super().__init__()
        if grid_size != 5 or spline_order != 3 or not enable_standalone_scale_spline or base_activation is not nn.SiLU:
            raise NotImplementedError("KANLinear: the HIP kernels are built for grid_size=5, spline_order=3, SiLU base, "
                                      "standalone spline scaler (the configuration KM-UNet uses)")
        self.in_features, self.out_features = in_features, out_features
        self.grid_size, self.spline_order = grid_size, spline_order
        h = (grid_range[1] - grid_range[0]) / grid_size
        knots = torch.arange(-spline_order, grid_size + spline_order + 1) * h + grid_range[0]
        self.register_buffer("grid", knots.expand(in_features, -1).contiguous())
        self.base_weight = nn.Parameter(torch.empty(out_features, in_features))
        self.spline_weight = nn.Parameter(torch.empty(out_features, in_features, grid_size + spline_order))
        self.spline_scaler = nn.Parameter(torch.empty(out_features, in_features))
        self.scale_noise, self.scale_base, self.scale_spline = scale_noise, scale_base, scale_spline
        self.grid_eps = grid_eps
        self.reset_parameters()

    def reset_parameters(self):
        # Same distributions as the reference (KANlayers.py:555-575); the spline coefficients are the
        # least-squares fit of uniform noise at the 6 interior knots, solved here in closed form per
        # feature with torch.linalg.lstsq on the (6 x 8) collocation matrix of the shared knot vector.
        nn.init.kaiming_uniform_(self.base_weight, a=math.sqrt(5) * self.scale_base)
        nn.init.kaiming_uniform_(self.spline_scaler, a=math.sqrt(5) * self.scale_spline)
        with torch.no_grad():
            g = self.grid[0].double()
            pts = g[self.spline_order:-self.spline_order]                       # 6 interior knots
            coll = _bspline_collocation(pts, g, self.spline_order)              # [6, 8]
            noise = (torch.rand(self.grid_size + 1, self.in_features * self.out_features, dtype=torch.float64) - 0.5) \
                * self.scale_noise / self.grid_size
            coef = torch.linalg.lstsq(coll, noise).solution                     # [8, in*out]
            self.spline_weight.copy_(coef.t().reshape(self.in_features, self.out_features, -1).permute(1, 0, 2))


    @property
    def scaled_spline_weight(self):
        """KANlayers.py:644-650 (enable_standalone_scale_spline is always on here)."""
        return self.spline_weight * self.spline_scaler.unsqueeze(-1)

    def regularization_loss(self, regularize_activation=1.0, regularize_entropy=1.0):
        """KANlayers.py:713-731: with a[o,i] = mean_k |spline_weight[o,i,k]|, returns
        regularize_activation * sum(a) + regularize_entropy * entropy(a / sum(a)).  Parameter-only, plain tensor ops."""
        a = self.spline_weight.abs().mean(dim=-1)
        total = a.sum()
        p = a / total
        return regularize_activation * total - regularize_entropy * torch.sum(p * torch.log(p))

    # ---- the row-wise form (KANlayers.py:577-660) in plain tensor arithmetic: needed by update_grid and, after it, by a layer whose
    # input features no longer share one knot vector.  NOT the accelerated path: KANConv2d runs on the HIP kernels whenever the grid
    # rows are identical (always, unless update_grid was called -- the reference never calls it).
    def b_splines(self, x):
        """[M, in] -> [M, in, grid_size + spline_order]: Cox-de Boor on each feature's own knots, half-open intervals (:577-610)."""
        g = self.grid
        x = x.unsqueeze(-1)
        b = ((x >= g[:, :-1]) & (x < g[:, 1:])).to(x.dtype)
        for k in range(1, self.spline_order + 1):
            b = (x - g[:, :-(k + 1)]) / (g[:, k:-1] - g[:, :-(k + 1)]) * b[:, :, :-1] \
                + (g[:, k + 1:] - x) / (g[:, k + 1:] - g[:, 1:-k]) * b[:, :, 1:]
        return b.contiguous()

    def curve2coeff(self, x, y):
        """Least-squares spline coefficients [out, in, coeff] of the curves through (x [M, in], y [M, in, out]) (:612-642)."""
        sol = torch.linalg.lstsq(self.b_splines(x).transpose(0, 1), y.transpose(0, 1)).solution      # [in, coeff, out]
        return sol.permute(2, 0, 1).contiguous()

    def forward(self, x):
        """y = SiLU(x) Wb^T + vec(B(x)) (Ws . scaler)^T on rows [M, in] (:652-660), stock tensor ops on x's device."""
        base = torch.nn.functional.linear(torch.nn.functional.silu(x), self.base_weight)
        spline = torch.nn.functional.linear(self.b_splines(x).view(x.size(0), -1), self.scaled_spline_weight.view(self.out_features, -1))
        return base + spline

    @torch.no_grad()
    def update_grid(self, x, margin=0.01):
        """KANlayers.py:662-709: re-fit every input feature's knot vector to the distribution of its column of x [M, in] (grid_eps
        blends a uniform grid over the column's range with its quantiles), then re-fit the spline coefficients so that the layer's
        spline outputs on x are preserved.  Afterwards the features no longer share one knot vector: KANConv2d then evaluates this
        layer through `forward` above (unfold + tensor ops) instead of the HIP kernels, which are built for the shared grid."""
        assert x.dim() == 2 and x.size(1) == self.in_features
        m = x.size(0)
        per_feature = torch.bmm(self.b_splines(x).permute(1, 0, 2), self.scaled_spline_weight.permute(1, 2, 0)).permute(1, 0, 2)   # [M, in, out]
        xs = torch.sort(x, dim=0)[0]
        adaptive = xs[torch.linspace(0, m - 1, self.grid_size + 1, dtype=torch.int64, device=x.device)]
        step = (xs[-1] - xs[0] + 2 * margin) / self.grid_size
        uniform = torch.arange(self.grid_size + 1, dtype=torch.float32, device=x.device).unsqueeze(1) * step + xs[0] - margin
        grid = self.grid_eps * uniform + (1 - self.grid_eps) * adaptive
        k = self.spline_order
        grid = torch.cat([grid[:1] - step * torch.arange(k, 0, -1, device=x.device).unsqueeze(1), grid,
                          grid[-1:] + step * torch.arange(1, k + 1, device=x.device).unsqueeze(1)], dim=0)
        self.grid.copy_(grid.T)
        self.spline_weight.data.copy_(self.curve2coeff(x, per_feature))


def _bspline_collocation(x, knots, order):
    x = x.unsqueeze(-1)
    b = ((x >= knots[:-1]) & (x < knots[1:])).to(x.dtype)
    for k in range(1, order + 1):
        b = (x - knots[:-(k + 1)]) / (knots[k:-1] - knots[:-(k + 1)]) * b[:, :-1] \
            + (knots[k + 1:] - x) / (knots[k + 1:] - knots[1:-k]) * b[:, 1:]
    return b


class KANConv2d(nn.Module):
    """convKAN/KANConv2Dlayers.py:5-37.  Only the 3x3 / stride 1 / padding 1 form KM-UNet uses is built."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        if (kernel_size, stride, padding) != (3, 1, 1):
            raise NotImplementedError("KANConv2d: HIP kernel is built for kernel_size=3, stride=1, padding=1")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.kanlayer = KANLinear(in_channels * kernel_size * kernel_size, out_channels)

    def forward(self, x, residual=None, relu=False):
        k = self.kanlayer
        if not _shared_grid(k.grid):
            # update_grid() gave every (channel, tap) feature its own knots: Phi can no longer be evaluated once per input element,
            # which is what the HIP kernels are built on -- the layer runs in its row-wise form (unfold + tensor ops on x's device).
            # No model of the reference reaches this state (update_grid is never called: SURVEY.md section 7).
            from .kan_variants import fold_rows, unfold_rows
            rows, shape = unfold_rows(x, self.kernel_size, self.stride, self.padding)
            y = fold_rows(k(rows), shape, self.out_channels)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        return ops.kan_conv2d(x, k.grid, k.base_weight, k.spline_weight, k.spline_scaler, residual, relu)


_GRID_SHARED = {}


def _shared_grid(grid):
    """True when every row of KANLinear.grid is the same knot vector (checked once per buffer version)."""
    key = (grid.data_ptr(), grid._version, tuple(grid.shape))
    v = _GRID_SHARED.get(key)
    if v is None:
        if len(_GRID_SHARED) > 256:
            _GRID_SHARED.clear()
        v = _GRID_SHARED[key] = bool((grid == grid[0:1]).all())
    return v


# ------------------------------------------------------------------ vim_block_init (K2)
class LayerNorm1D(nn.Module):
    def __init__(self, num_channels, eps=1e-5, affine=True):
        super().__init__()
        assert affine
        self.num_channels, self.eps = num_channels, eps
        self.weight = nn.Parameter(torch.ones(1, num_channels, 1))
        self.bias = nn.Parameter(torch.zeros(1, num_channels, 1))

    def forward(self, x):
        return ops.layernorm1d(x, self.weight, self.bias, self.eps)


class ConvLayer1D(nn.Module):
    """vim_utils_init.py:92-119 restricted to what KM-UNet instantiates (1x1, bias-free, optional BN/act)."""

    def __init__(self, in_dim, out_dim, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, norm=nn.BatchNorm1d,
                 act_layer=nn.ReLU, bn_weight_init=1):
        super().__init__()
        self.conv = nn.Conv1d(in_dim, out_dim, kernel_size, stride, padding, dilation, groups, bias=False)
        self.norm = norm(num_features=out_dim) if norm else None
        self.act = act_layer() if act_layer else None
        if self.norm:
            nn.init.constant_(self.norm.weight, bn_weight_init)
            nn.init.constant_(self.norm.bias, 0)

    def forward(self, x):
        x = self.conv(x)
        if self.norm:
            x = self.norm(x)
        return self.act(x) if self.act else x


class ConvLayer2D(nn.Module):
    """vim_utils_init.py:62-89."""

    def __init__(self, in_dim, out_dim, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, norm=nn.BatchNorm2d,
                 act_layer=nn.ReLU, bn_weight_init=1):
        super().__init__()
        self.conv = nn.Conv2d(in_dim, out_dim, kernel_size, stride, padding, dilation, groups, bias=False)
        self.norm = norm(num_features=out_dim) if norm else None
        self.act = act_layer() if act_layer else None
        if self.norm:
            nn.init.constant_(self.norm.weight, bn_weight_init)
            nn.init.constant_(self.norm.bias, 0)

    def conv_only(self, x):
        c = self.conv
        if c.groups == c.in_channels == c.out_channels and c.kernel_size == (3, 3) and c.stride == (1, 1) \
                and c.padding == (1, 1) and c.dilation == (1, 1) and "dwconv" not in _TORCH_GLUE:
            return ops.dwconv3x3(x, c.weight, c.bias)       # HIP stencil instead of MIOpen's naive fallback
        if _is_pointwise(c):
            return conv1x1(x, c)
        return c(x)

    def forward(self, x):
        x = self.conv_only(x)
        if isinstance(self.norm, nn.BatchNorm2d) and (self.act is None or isinstance(self.act, nn.ReLU)) \
                and "bn_blend" not in _TORCH_GLUE:
            return ops.bn_blend(x, None, self.norm, None, 0, relu=self.act is not None)
        if self.norm:
            x = self.norm(x)
        return self.act(x) if self.act else x


class HSMSSD(nn.Module):
    """vim_block_init/efficient_vim_init.py:14-61; forward = one fused HIP op returning (y, h)."""

    def __init__(self, d_model, ssd_expand=1, A_init_range=(1, 16), state_dim=64):
        super().__init__()
        if ssd_expand != 1:
            raise NotImplementedError("HSMSSD: ssd_expand != 1 is not built (KM-UNet uses 1)")
        self.ssd_expand, self.d_inner, self.state_dim = ssd_expand, d_model, state_dim
        self.BCdt_proj = ConvLayer1D(d_model, 3 * state_dim, 1, norm=None, act_layer=None)
        self.dw = ConvLayer2D(3 * state_dim, 3 * state_dim, 3, 1, 1, groups=3 * state_dim, norm=None, act_layer=None)
        self.hz_proj = ConvLayer1D(d_model, 2 * d_model, 1, norm=None, act_layer=None)
        self.out_proj = ConvLayer1D(d_model, d_model, 1, norm=None, act_layer=None)
        self.A = nn.Parameter(torch.empty(state_dim).uniform_(*A_init_range))
        self.D = nn.Parameter(torch.ones(1))
        self.D._no_weight_decay = True

    def forward(self, x):
        return ops.hsmssd(x, self.BCdt_proj.conv.weight, self.dw.conv.weight, self.hz_proj.conv.weight,
                          self.out_proj.conv.weight, self.A, self.D)


class FFN(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.fc1 = ConvLayer2D(in_dim, dim, 1)
        self.fc2 = ConvLayer2D(dim, in_dim, 1, act_layer=None, bn_weight_init=0)

    def forward(self, x):
        return self.fc2(self.fc1(x))


class EfficientViMBlock(nn.Module):
    """efficient_vim_init.py:64-97."""

    def __init__(self, dim, mlp_ratio=4.0, ssd_expand=1, state_dim=64):
        super().__init__()
        self.dim, self.mlp_ratio = dim, mlp_ratio
        self.mixer = HSMSSD(d_model=dim, ssd_expand=ssd_expand, state_dim=state_dim)
        self.norm = LayerNorm1D(dim)
        self.dwconv1 = ConvLayer2D(dim, dim, 3, padding=1, groups=dim, bn_weight_init=0, act_layer=None)
        self.dwconv2 = ConvLayer2D(dim, dim, 3, padding=1, groups=dim, bn_weight_init=0, act_layer=None)
        self.ffn = FFN(in_dim=dim, dim=int(dim * mlp_ratio))
        self.alpha = nn.Parameter(1e-4 * torch.ones(4, dim))

    def forward(self, x):
        # x <- (1-a_k) x + a_k f_k(x), a = sigmoid(alpha): each blend (and the BatchNorm / ReLU in front of it)
        # is one fused HIP op (csrc/bn_blend.hip)
        if "bn_blend" in _TORCH_GLUE:
            a = torch.sigmoid(self.alpha).view(4, -1, 1, 1)
            x = torch.lerp(x, self.dwconv1(x), a[0])
            y, _ = self.mixer(self.norm(x.flatten(2)))
            x = torch.lerp(x, y, a[1])
            x = torch.lerp(x, self.dwconv2(x), a[2])
            return torch.lerp(x, self.ffn(x), a[3])
        a0, a1, a2, a3 = self.alpha.unbind(0)     # one stack() in backward instead of 4 x (zeros + add)
        b, c, hh, ww = x.shape
        if "evim_composite" not in _TORCH_GLUE and "dwconv" not in _TORCH_GLUE and "pwconv" not in _TORCH_GLUE \
                and ops.pwconv_supported(c, self.ffn.fc1.conv.out_channels, hh * ww):
            # each stage as ONE autograd node: its backward folds the blend partner's gradient into the branch's last kernel
            x = ops.dw_bn_blend(x, self.dwconv1.conv, self.dwconv1.norm, a0)
            # LayerNorm1D + HSMSSD as one node of two launches (csrc/hsmssd_v2.inc).  x feeds the norm AND the blend: the blend takes an
            # alias whose gradient the LayerNorm backward kernel adds in
            mx = self.mixer
            y, _, xa = ops.mixer_ln(x.flatten(2), self.norm.weight, self.norm.bias, self.norm.eps, mx.BCdt_proj.conv.weight, mx.dw.conv.weight,
                                    mx.hz_proj.conv.weight, mx.out_proj.conv.weight, mx.A, mx.D, alias=True)
            x = ops.bn_blend(y, xa.view_as(x), None, a1)
            x = ops.dw_bn_blend(x, self.dwconv2.conv, self.dwconv2.norm, a2)
            return ops.ffn_blend(x, self.ffn.fc1, self.ffn.fc2, a3)
        x = ops.bn_blend(self.dwconv1.conv_only(x), x, self.dwconv1.norm, a0)
        y, _ = self.mixer(self.norm(x.flatten(2)))
        x = ops.bn_blend(y, x, None, a1)
        x = ops.bn_blend(self.dwconv2.conv_only(x), x, self.dwconv2.norm, a2)
        h = self.ffn.fc1(x)
        return ops.bn_blend(self.ffn.fc2.conv_only(h), x, self.ffn.fc2.norm, a3)


# ------------------------------------------------------------------ DySample (K3)
class DySample(nn.Module):
    """DySample_md.py:20-81, style 'lp' without dyscope (the configuration KM-UNet builds)."""

    def __init__(self, in_channels, scale=2, style="lp", groups=4, dyscope=False):
        super().__init__()
        if style != "lp" or dyscope or scale != 2 or groups != 4:
            raise NotImplementedError("DySample: HIP kernel is built for scale=2, style='lp', groups=4, dyscope=False")
        self.scale, self.style, self.groups = scale, style, groups
        self.offset = nn.Conv2d(in_channels, 2 * groups * scale ** 2, 1)
        nn.init.normal_(self.offset.weight, 0, 0.001)
        nn.init.constant_(self.offset.bias, 0)
        h = torch.arange((-scale + 1) / 2, (scale - 1) / 2 + 1) / scale
        self.register_buffer("init_pos", torch.stack(torch.meshgrid([h, h], indexing="ij")).transpose(1, 2)
                             .repeat(1, groups, 1).reshape(1, -1, 1, 1))

    def forward(self, x):
        return ops.dysample_lp(x, conv1x1(x, self.offset), self.init_pos)


# ------------------------------------------------------------------ DAGEM (K4)
class DeformConv2d(nn.Module):
    """Parameters of torchvision.ops.DeformConv2d(in, out, 3, padding=1): weight, bias."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True):
        super().__init__()
        if (kernel_size, stride, padding) != (3, 1, 1):
            raise NotImplementedError("DeformConv2d: HIP kernel is built for 3x3 / stride 1 / padding 1")
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_channels * 9)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, offset):
        return ops.deform_conv2d(x, offset, self.weight, self.bias)


class _SkinnyLinearFn(torch.autograd.Function):
    """nn.Linear(k, 1) on [N, k] with N ~ 1e5 (DAGEM_md.py:18-22, :33-37).  Forward and input gradient are ATen's; the weight
    gradient dy^T x is a [1, N] x [N, k] product that hipBLASLt serves with a 16x16x512 macro-tile kernel on ONE workgroup
    (113 us at N = 131072, rocprof) -- as a product and a column sum it is two streaming launches (~12 us)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)      # fp32 inside whatever autocast says outside
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.float()
        return dy @ w, (x * dy).sum(0, keepdim=True), dy.sum(0)


def _bn_relu_rows(bn, yt):
    """ReLU(BatchNorm1d(y)) for y given CHANNEL-MAJOR as yt [o, N] (row n of the reference's [N, o] activations = column n):
    per-feature statistics over N are the NCHW BatchNorm kernel with B = 1, C = o, HW = N (csrc/bn_blend.hip: statistics +
    apply-with-ReLU, 2 launches each way instead of ATen's batch_norm + relu chains).  Returns [o, N]."""
    o = yt.shape[0]
    tap, ops.RELU_TAP = ops.RELU_TAP, None          # the parity tooling wants the mask in the reference's [N, o] layout
    try:
        out = ops.bn_blend(yt.reshape(1, o, -1), None, bn, None, relu=True).view(o, -1)
    finally:
        ops.RELU_TAP = tap
    if tap is not None:
        tap.append((out.detach().t() > 0).cpu())
    return out


def _mlp_rows(seq, x):
    """nn.Sequential(Linear(k, o), BatchNorm1d(o), ReLU) on rows x [N, k] (DAGEM_md.py:14-38), result CHANNEL-MAJOR [o, N]."""
    lin, bn = seq[0], seq[1]
    if lin.out_features == 1:
        yt = _SkinnyLinearFn.apply(x, lin.weight, lin.bias).view(1, -1)
    else:
        yt = torch.addmm(lin.bias.unsqueeze(1), lin.weight, x.t())       # W x^T + b: [o, N] without a transpose pass
    return _bn_relu_rows(bn, yt)


def _conv_bn_relu(seq, t):
    """The same Sequential(Linear(k, o), BatchNorm1d(o), ReLU) applied to the channel axis of t [B, k, ...]: a Linear over the
    channels-last rows IS a 1x1 convolution, and BatchNorm1d over those rows is BatchNorm2d -- so the reference's
    cat -> permute -> reshape -> Linear -> ... -> view -> permute round trips (DAGEM_md.py:66-69, :74-81) disappear: NCHW in, NCHW
    out, on the pointwise-conv and BatchNorm kernels.  The ReLU mask is reported in the reference's [rows, o] layout."""
    lin, bn = seq[0], seq[1]
    y = ops.pwconv(t.reshape(t.shape[0], t.shape[1], -1, 1), lin.weight, lin.bias)
    tap, ops.RELU_TAP = ops.RELU_TAP, None
    try:
        out = ops.bn_blend(y, None, bn, None, relu=True)
    finally:
        ops.RELU_TAP = tap
    if tap is not None:
        tap.append((out.detach().flatten(2).permute(0, 2, 1).reshape(-1, out.shape[1]) > 0).cpu())
    return out


def _mlp(seq, x):
    lin = seq[0]
    y = _SkinnyLinearFn.apply(x, lin.weight, lin.bias) if (lin.out_features == 1 and x.is_cuda) else lin(x)
    return seq[2](seq[1](y))


class DAGEM(nn.Module):
    """DAGEM_md.py:7-111: graph-edge bridge; the deformable conv is the HIP kernel, the small MLPs are glue."""

    def __init__(self, sync_bn=False, input_channels=256):
        super().__init__()
        c = self.input_channels = input_channels
        mlp = lambda i, o: nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU(inplace=True))
        self.edge_aggregation_func = mlp(4, 1)
        self.vertex_update_func = mlp(2 * c, c // 2)
        self.edge_update_func = mlp(2 * c, c // 2)
        self.update_edge_reduce_func = mlp(4, 1)
        self.offset_conv = nn.Conv2d(c, 18, 3, padding=1)
        self.deform_conv = DeformConv2d(c, c, 3, padding=1)
        self.final_aggregation_layer = nn.Sequential(nn.Conv2d(c + c // 2, c, 1, bias=False), nn.BatchNorm2d(c),
                                                     nn.ReLU(inplace=True))

    def forward(self, x):
        b, c, h, w = x.shape
        if x.is_cuda and "dagem" not in _TORCH_GLUE and _DAGEM_FUSED:
            bns = [s_[1] for s_ in (self.edge_aggregation_func, self.vertex_update_func, self.edge_update_func,
                                    self.update_edge_reduce_func)] + [self.final_aggregation_layer[1]]
            if ops.dagem_supported(x, bns):
                # one launch per BatchNorm boundary (csrc/dagem_fused.hip); the residual of :101 is added where the concatenation is read
                return ops.dagem_glue(x, self.deform_conv(x, conv3x3(x, self.offset_conv)), self)
        if x.is_cuda and "dagem" not in _TORCH_GLUE:
            # same arithmetic, fewer launches: edge products by one gather kernel, Linear outputs kept channel-major so that
            # BatchNorm1d + ReLU run on the NCHW BatchNorm kernels
            edge = ops.dagem_edges(x)                                                     # [B,C,H,W,4]
            agg = _mlp_rows(self.edge_aggregation_func, edge.reshape(-1, 4)).view(b, c, h, w)
            ef = torch.cat((x.unsqueeze(-1).expand_as(edge), edge), 1)                    # [B,2C,H,W,4]
            if ops.pwconv_supported(2 * c, c // 2, h * w):
                vert = _conv_bn_relu(self.vertex_update_func, torch.cat((x, agg), 1)).view(b, c // 2, h, w)
                ue = _conv_bn_relu(self.edge_update_func, ef)                             # [B,C/2,(H,W,4)]: rows of 4 already contiguous
                ue = _mlp_rows(self.update_edge_reduce_func, ue.reshape(-1, 4)).view(b, c // 2, h, w)
            else:
                vert = _mlp_rows(self.vertex_update_func, torch.cat((x, agg), 1).permute(0, 2, 3, 1).reshape(-1, 2 * c))
                vert = vert.view(c // 2, b, h, w).permute(1, 0, 2, 3)                      # [c/2, (b,h,w)] -> [B,c/2,H,W] (view)
                ue = _mlp_rows(self.edge_update_func, ef.permute(0, 2, 3, 4, 1).reshape(-1, 2 * c))
                ue = ue.view(c // 2, b, h, w, 4).permute(1, 0, 2, 3, 4).reshape(-1, 4)
                ue = _mlp_rows(self.update_edge_reduce_func, ue).view(b, c // 2, h, w)
        else:
            nb = torch.stack((x.roll(1, 2), x.roll(-1, 2), x.roll(1, 3), x.roll(-1, 3)), dim=-1)
            edge = nb * x.unsqueeze(-1)                                                   # [B,C,H,W,4]
            agg = _mlp(self.edge_aggregation_func, edge.reshape(-1, 4)).view(b, c, h, w)
            vert = self.vertex_update_func(torch.cat((x, agg), 1).permute(0, 2, 3, 1).reshape(-1, 2 * c))
            vert = vert.view(b, h, w, c // 2).permute(0, 3, 1, 2)
            ef = torch.cat((x.unsqueeze(-1).expand_as(edge), edge), 1).permute(0, 2, 3, 4, 1).reshape(-1, 2 * c)
            ue = self.edge_update_func(ef).view(b, h, w, 4, c // 2).permute(0, 4, 1, 2, 3).reshape(-1, 4)
            ue = _mlp(self.update_edge_reduce_func, ue).view(b, c // 2, h, w)
        deformed = self.deform_conv(x, conv3x3(x, self.offset_conv)) + x
        fa = self.final_aggregation_layer
        z = conv1x1(torch.cat((deformed, vert * ue), 1), fa[0])
        if x.is_cuda and "dagem" not in _TORCH_GLUE and "bn_blend" not in _TORCH_GLUE:
            return ops.bn_blend(z, None, fa[1], None, relu=True)        # BatchNorm2d + ReLU: statistics + apply, 2 launches each way
        return fa[2](fa[1](z))


# ------------------------------------------------------------------ WPL/iwp.py (glue)
class _HaarDWT(nn.Module):
    """WPL/iwp.py:47-113 as a 2x2 stencil.  LL = (a+b+c+d)/2 etc.; the reference's high-pass matrix has
    an all-zero last row (iwp.py:79), i.e. the last row of HL/HH and the last column of LH/HH are zero."""

    def forward(self, x):
        a, b = x[..., 0::2, 0::2], x[..., 0::2, 1::2]
        c, d = x[..., 1::2, 0::2], x[..., 1::2, 1::2]
        ll = (a + b + c + d) * 0.5
        lh = (a - b + c - d) * 0.5      # low over rows, high over columns
        hl = (a + b - c - d) * 0.5      # high over rows, low over columns
        hh = (a - b - c + d) * 0.5
        hq, wq = ll.shape[-2:]
        rmask = torch.ones(hq, 1, device=x.device, dtype=x.dtype)
        rmask[-1] = 0
        cmask = torch.ones(1, wq, device=x.device, dtype=x.dtype)
        cmask[:, -1] = 0
        return ll, lh * cmask, hl * rmask, hh * (rmask * cmask)


class IntelligentWaveletPoolingModule(nn.Module):
    """WPL/iwp.py:116-132.  nn.Softmax2d over the one-channel attention map is identically 1."""

    def __init__(self, in_channels, wavename="haar"):
        super().__init__()
        if wavename != "haar":
            raise NotImplementedError("only the haar wavelet is restated")
        self.dwt = _HaarDWT()
        self.high_freq_conv = nn.Conv2d(3 * in_channels, 1, 1)
        self.softmax = nn.Softmax2d()
        self.fusion_conv = nn.Conv2d(in_channels + 1, in_channels, 1)

    def forward(self, x):
        if "iwp" not in _TORCH_GLUE and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            # DWT + (identically-1) attention + channel mean + concat: one HIP stencil (csrc/iwp.hip)
            hf, fc = self.high_freq_conv, self.fusion_conv
            c, co = x.shape[1], fc.out_channels
            ct = (c + 1 + 15) // 16 * 16
            if "pwconv" not in _TORCH_GLUE and ops.pwconv_supported(ct, co, (x.shape[2] // 2) * (x.shape[3] // 2)):
                # C+1 input channels padded to a multiple of 16 (zero channels x zero weight columns) => pointwise-conv kernels
                # instead of the strided-batched GEMM path (whose weight gradient ran 113 us per module on rocBLAS)
                w = F.pad(fc.weight, (0, 0, 0, 0, 0, ct - (c + 1)))
                return ops.pwconv(ops.iwp_front(x, hf.weight, hf.bias, ct), w, fc.bias)
            return conv1x1(ops.iwp_front(x, hf.weight, hf.bias), fc)
        ll, lh, hl, hh = self.dwt(x)
        high = torch.cat([lh, hl, hh], dim=1)
        high = high * self.softmax(conv1x1(high, self.high_freq_conv))
        return conv1x1(torch.cat([ll, high.mean(dim=1, keepdim=True)], dim=1), self.fusion_conv)
