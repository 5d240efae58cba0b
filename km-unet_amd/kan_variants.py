"""The eight alternative KAN convolutions of convKAN/KANConv2Dlayers.py:40-293 as pass-through PyTorch modules.

KM_UNetV3 builds `KANConv2d` only -- the alternatives are imported and commented out at KM_UNetV3_SH.py:28-32 -- so none of this is
on the accelerated path and nothing here launches a HIP kernel: every class is stock tensor arithmetic that runs wherever its tensors
live.  They exist so that `from convKAN.KANConv2Dlayers import *` gives working layers with the reference's constructor signatures,
attribute names (hence state_dict keys) and numerics; tests/golden/kanvar_*.npz pins each one against the reference on CPU.

Every conv wrapper is the same three steps (KANConv2Dlayers.py:50-68 and its seven repetitions): unfold k x k patches into rows
[B*L, Cin*k*k] (feature index c*k*k + ky*k + kx), apply the row-wise KAN layer, fold the [B*L, Cout] rows back to NCHW.
The layers, by the formula each evaluates per input feature x_i (KANlayers.py line ranges in the class docstrings):

  Chebyshev   y_o = sum_i sum_d c[i,o,d] T_d(clamp(x_i, -1, 1)),  T_d(t) = cos(d arccos t)
  FastKAN     y = W_s vec(exp(-((LN(x)_i - g_k) / h)^2)) + W_b SiLU(x) + b        (Gaussian RBF on a uniform grid g, spacing h)
  Gram        y = SiLU(LN(sum_i sum_d w[i,o,d] SiLU(P_d(tanh x_i)) + W_b SiLU(x))),  P_0 = 1, P_1 = t, P_d = t P_{d-1} - beta(d-1, d) P_{d-2}
  Wavelet     y = BN(sum_i w[o,i] psi((x_i - tau[o,i]) / s[o,i]) + W_1 SiLU(x)),  psi = Mexican hat / Morlet / DoG / Meyer / Shannon
  Jacobi      y = SiLU(LN(sum_i sum_d c[i,o,d] J_d^{(a,b)}(tanh x_i) + W_b SiLU(x))),  three-term recurrence of the Jacobi polynomials
  ReLU-KAN    phi_k(x_i) = (r relu(x_i - lo[i,k]) relu(hi[i,k] - x_i))^2, y = Conv2d(1, out, (g + k, in))(phi)   (a dense contraction)
  FasterKAN   y = W_s vec(1 - tanh^2((LN(x)_i - g_k)))     (the reflectional switch; its custom backward scales by 1 / denominator)
  RBF         y = vec(exp(-((x_i - g_k) / h)^2)) W_s + W_b SiLU(x) + b
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- row-wise layers
class ChebyKANLayer(nn.Module):
    """KANlayers.py:123-156."""

    def __init__(self, input_dim, output_dim, degree):
        super().__init__()
        self.inputdim, self.outdim, self.degree = input_dim, output_dim, degree
        self.cheby_coeffs = nn.Parameter(torch.empty(input_dim, output_dim, degree + 1))
        nn.init.normal_(self.cheby_coeffs, mean=0.0, std=1 / (input_dim * (degree + 1)))
        self.register_buffer("arange", torch.arange(0, degree + 1, 1))

    def forward(self, x):
        t = torch.clamp(x, -1.0, 1.0).reshape(-1, self.inputdim, 1)
        basis = torch.cos(torch.acos(t) * self.arange)                        # [M, in, degree + 1]
        return torch.einsum("bid,iod->bo", basis, self.cheby_coeffs).view(-1, self.outdim)


class RadialBasisFunction(nn.Module):
    """convKAN/utils.py:9-24."""

    def __init__(self, grid_min=-2.0, grid_max=2.0, num_grids=8, denominator=None):
        super().__init__()
        self.grid = nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=False)
        self.denominator = denominator or (grid_max - grid_min) / (num_grids - 1)

    def forward(self, x):
        return torch.exp(-((x[..., None] - self.grid) / self.denominator) ** 2)


class SplineLinear(nn.Linear):
    """KANlayers.py:80-86: bias-free linear map with a truncated-normal start."""

    def __init__(self, in_features, out_features, init_scale=0.1, **kw):
        self.init_scale = init_scale
        super().__init__(in_features, out_features, bias=False, **kw)

    def reset_parameters(self):
        nn.init.trunc_normal_(self.weight, mean=0, std=self.init_scale)


class FastKANLayer(nn.Module):
    """KANlayers.py:89-119."""

    def __init__(self, input_dim, output_dim, grid_min=-2.0, grid_max=2.0, num_grids=8, use_base_update=True, base_activation=nn.SiLU,
                 spline_weight_init_scale=0.1):
        super().__init__()
        self.layernorm = nn.LayerNorm(input_dim)
        self.rbf = RadialBasisFunction(grid_min, grid_max, num_grids)
        self.spline_linear = SplineLinear(input_dim * num_grids, output_dim, spline_weight_init_scale)
        self.use_base_update = use_base_update
        if use_base_update:
            self.base_activation = base_activation()
            self.base_linear = nn.Linear(input_dim, output_dim)

    def forward(self, x, time_benchmark=False):
        phi = self.rbf(x if time_benchmark else self.layernorm(x))
        out = self.spline_linear(phi.flatten(-2))
        if self.use_base_update:
            out = out + self.base_linear(self.base_activation(x))
        return out


class GRAMLayer(nn.Module):
    """KANlayers.py:159-230."""

    def __init__(self, in_channels, out_channels, degree=3, act=nn.SiLU):
        super().__init__()
        self.in_channels, self.out_channels, self.degrees = in_channels, out_channels, degree
        self.act = act()
        self.norm = nn.LayerNorm(out_channels, dtype=torch.float32)
        self.beta_weights = nn.Parameter(torch.zeros(degree + 1, dtype=torch.float32))
        self.grams_basis_weights = nn.Parameter(torch.zeros(in_channels, out_channels, degree + 1, dtype=torch.float32))
        self.base_weights = nn.Parameter(torch.zeros(out_channels, in_channels, dtype=torch.float32))
        self.init_weights()

    def init_weights(self):
        nn.init.normal_(self.beta_weights, mean=0.0, std=1.0 / (self.in_channels * (self.degrees + 1.0)))
        nn.init.xavier_uniform_(self.grams_basis_weights)
        nn.init.xavier_uniform_(self.base_weights)

    def beta(self, n, m):
        return ((m + n) * (m - n) * n ** 2) / (m ** 2 / (4.0 * n ** 2 - 1.0)) * self.beta_weights[n]

    def gram_poly(self, t, degree):
        p = [torch.ones_like(t)]
        if degree >= 1:
            p.append(t)
        for d in range(2, degree + 1):
            p.append(t * p[-1] - self.beta(d - 1, d) * p[-2])
        return torch.stack(p, dim=-1)

    def forward(self, x):
        base = F.linear(self.act(x), self.base_weights)
        basis = self.act(self.gram_poly(torch.tanh(x), self.degrees))        # [M, in, degree + 1]
        y = torch.einsum("bld,lod->bo", basis, self.grams_basis_weights)
        return self.act(self.norm(y + base)).view(-1, self.out_channels)


class WavKANLayer(nn.Module):
    """KANlayers.py:233-321."""

    def __init__(self, in_features, out_features, wavelet_type="mexican_hat"):
        super().__init__()
        self.in_features, self.out_features, self.wavelet_type = in_features, out_features, wavelet_type
        self.scale = nn.Parameter(torch.ones(out_features, in_features))
        self.translation = nn.Parameter(torch.zeros(out_features, in_features))
        self.weight1 = nn.Parameter(torch.empty(out_features, in_features))
        self.wavelet_weights = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.wavelet_weights, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.weight1, a=math.sqrt(5))
        self.base_activation = nn.SiLU()
        self.bn = nn.BatchNorm1d(out_features)

    def wavelet_transform(self, x):
        u = ((x.unsqueeze(1) if x.dim() == 2 else x) - self.translation) / self.scale          # [M, out, in]
        kind = self.wavelet_type
        if kind == "mexican_hat":
            psi = (2 / (math.sqrt(3) * math.pi ** 0.25)) * (u ** 2 - 1) * torch.exp(-0.5 * u ** 2)
        elif kind == "morlet":
            psi = torch.exp(-0.5 * u ** 2) * torch.cos(5.0 * u)
        elif kind == "dog":
            psi = -u * torch.exp(-0.5 * u ** 2)
        elif kind == "meyer":
            v = u.abs()
            t = 2 * v - 1
            nu = t ** 4 * (35 - 84 * t + 70 * t ** 2 - 20 * t ** 3)
            aux = torch.where(v <= 0.5, torch.ones_like(v), torch.where(v >= 1, torch.zeros_like(v), torch.cos(math.pi / 2 * nu)))
            psi = torch.sin(math.pi * v) * aux
        elif kind == "shannon":
            win = torch.hamming_window(u.size(-1), periodic=False, dtype=u.dtype, device=u.device)
            psi = torch.sinc(u / math.pi) * win
        else:
            raise ValueError("Unsupported wavelet type")
        return (psi * self.wavelet_weights).sum(dim=2)

    def forward(self, x):
        return self.bn(self.wavelet_transform(x) + F.linear(self.base_activation(x), self.weight1))


class JacobiKANLayer(nn.Module):
    """KANlayers.py:324-367."""

    def __init__(self, input_dim, output_dim, degree, a=1.0, b=1.0, act=nn.SiLU):
        super().__init__()
        self.inputdim, self.outdim, self.a, self.b, self.degree = input_dim, output_dim, a, b, degree
        self.act = act()
        self.norm = nn.LayerNorm(output_dim, dtype=torch.float32)
        self.base_weights = nn.Parameter(torch.zeros(output_dim, input_dim, dtype=torch.float32))
        self.jacobi_coeffs = nn.Parameter(torch.empty(input_dim, output_dim, degree + 1))
        nn.init.normal_(self.jacobi_coeffs, mean=0.0, std=1 / (input_dim * (degree + 1)))
        nn.init.xavier_uniform_(self.base_weights)

    def forward(self, x):
        x = x.reshape(-1, self.inputdim)
        base = F.linear(self.act(x), self.base_weights)
        t = torch.tanh(x)
        a, b = self.a, self.b
        p = [torch.ones_like(t)]
        if self.degree > 0:
            p.append(((a - b) + (a + b + 2) * t) / 2)
        for i in range(2, self.degree + 1):
            k0 = (2 * i + a + b) * (2 * i + a + b - 1) / (2 * i * (i + a + b))
            k1 = (2 * i + a + b - 1) * (a * a - b * b) / (2 * i * (i + a + b) * (2 * i + a + b - 2))
            k2 = (i + a - 1) * (i + b - 1) * (2 * i + a + b) / (i * (i + a + b) * (2 * i + a + b - 2))
            p.append((k0 * t + k1) * p[-1] - k2 * p[-2])
        y = torch.einsum("bid,iod->bo", torch.stack(p, dim=-1), self.jacobi_coeffs).view(-1, self.outdim)
        return self.act(self.norm(y + base))


class ReLUKANLayer(nn.Module):
    """KANlayers.py:372-398."""

    def __init__(self, input_size, g, k, output_size, train_ab=True):
        super().__init__()
        self.g, self.k, self.r = g, k, 4 * g * g / ((k + 1) * (k + 1))
        self.input_size, self.output_size = input_size, output_size
        lo = torch.arange(-k, g, dtype=torch.float32) / g
        self.phase_low = nn.Parameter(lo.repeat(input_size, 1), requires_grad=train_ab)
        self.phase_height = nn.Parameter((lo + (k + 1) / g).repeat(input_size, 1), requires_grad=train_ab)
        self.equal_size_conv = nn.Conv2d(1, output_size, (g + k, input_size))

    def forward(self, x):
        xe = x.unsqueeze(2)
        phi = (torch.relu(xe - self.phase_low) * torch.relu(self.phase_height - xe) * self.r) ** 2      # [M, in, g + k]
        # the reference re-reads this buffer as [M, 1, g + k, in] WITHOUT transposing it (:393): kept, it is part of the layer
        return self.equal_size_conv(phi.reshape(len(phi), 1, self.g + self.k, self.input_size)).reshape(len(phi), self.output_size)


class _SwitchFn(torch.autograd.Function):
    """convKAN/utils.py:27-91: 1 - tanh^2(x - g); the backward multiplies by the inverse denominator although the forward does not
    divide by it (reference behaviour, kept).  grid / denominator gradients as the reference's (only when trainable)."""

    @staticmethod
    def forward(ctx, x, grid, inv_denominator, train_grid, train_inv):
        d = x[..., None] - grid
        th = torch.tanh(d)
        sech2 = 1 - th * th
        ctx.save_for_backward(th, sech2, d, inv_denominator)
        ctx.flags = (bool(train_grid), bool(train_inv))
        return sech2

    @staticmethod
    def backward(ctx, g):
        th, sech2, d, inv = ctx.saved_tensors
        gx = (-2 * th * sech2 * g).sum(dim=-1) * inv
        ggrid = -inv * g.sum(dim=0).sum(dim=0) if ctx.flags[0] else None
        ginv = (g * d).sum() if ctx.flags[1] else None
        return gx, ggrid, ginv, None, None


class ReflectionalSwitchFunction(nn.Module):
    """convKAN/utils.py:93-116."""

    def __init__(self, grid_min=-1.2, grid_max=0.2, num_grids=8, exponent=2, inv_denominator=0.5, train_grid=False,
                 train_inv_denominator=False):
        super().__init__()
        self.train_grid = torch.tensor(train_grid, dtype=torch.bool)
        self.train_inv_denominator = torch.tensor(train_inv_denominator, dtype=torch.bool)
        self.grid = nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=train_grid)
        self.inv_denominator = nn.Parameter(torch.tensor(inv_denominator, dtype=torch.float32), requires_grad=train_inv_denominator)

    def forward(self, x):
        return _SwitchFn.apply(x, self.grid, self.inv_denominator, self.train_grid, self.train_inv_denominator)


class SplineLinear_fstr(nn.Linear):
    """KANlayers.py:402-408: bias-free, Xavier-uniform start."""

    def __init__(self, in_features, out_features, init_scale=0.1, **kw):
        self.init_scale = init_scale
        super().__init__(in_features, out_features, bias=False, **kw)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)


class FasterKANLayer(nn.Module):
    """KANlayers.py:411-458."""

    def __init__(self, input_dim, output_dim, grid_min=-1.2, grid_max=0.2, num_grids=8, exponent=2, inv_denominator=0.5, train_grid=False,
                 train_inv_denominator=False, base_activation=F.silu, spline_weight_init_scale=0.667):
        super().__init__()
        self.layernorm = nn.LayerNorm(input_dim)
        self.rbf = ReflectionalSwitchFunction(grid_min, grid_max, num_grids, exponent, inv_denominator, train_grid, train_inv_denominator)
        self.spline_linear = SplineLinear_fstr(input_dim * num_grids, output_dim, spline_weight_init_scale)

    def forward(self, x):
        x = self.layernorm(x)
        return self.spline_linear(self.rbf(x).view(x.shape[0], -1))


class RBFLinear(nn.Module):
    """KANlayers.py:461-473."""

    def __init__(self, in_features, out_features, grid_min=-2.0, grid_max=2.0, num_grids=8, spline_weight_init_scale=0.1):
        super().__init__()
        self.grid_min, self.grid_max, self.num_grids = grid_min, grid_max, num_grids
        self.grid = nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=False)
        self.spline_weight = nn.Parameter(torch.randn(in_features * num_grids, out_features) * spline_weight_init_scale)

    def forward(self, x):
        h = (self.grid_max - self.grid_min) / (self.num_grids - 1)
        phi = torch.exp(-((x.unsqueeze(-1) - self.grid) / h) ** 2)
        return phi.reshape(phi.size(0), -1).matmul(self.spline_weight)


class RBFKANLayer(nn.Module):
    """KANlayers.py:476-492."""

    def __init__(self, input_dim, output_dim, grid_min=-2.0, grid_max=2.0, num_grids=8, use_base_update=True, base_activation=nn.SiLU(),
                 spline_weight_init_scale=0.1):
        super().__init__()
        self.input_dim, self.output_dim, self.use_base_update = input_dim, output_dim, use_base_update
        self.base_activation, self.spline_weight_init_scale = base_activation, spline_weight_init_scale
        self.rbf_linear = RBFLinear(input_dim, output_dim, grid_min, grid_max, num_grids, spline_weight_init_scale)
        self.base_linear = nn.Linear(input_dim, output_dim) if use_base_update else None

    def forward(self, x):
        out = self.rbf_linear(x)
        if self.use_base_update:
            out = out + self.base_linear(self.base_activation(x))
        return out


# ---------------------------------------------------------------------------------------------- conv wrappers
def unfold_rows(x, kernel_size, stride, padding):
    """[B, C, H, W] -> ([B*L, C*k*k] rows in F.unfold's feature order, (B, Ho, Wo))."""
    b, _, h, w = x.shape
    rows = F.unfold(x, kernel_size=kernel_size, stride=stride, padding=padding).transpose(1, 2)
    ho = (h + 2 * padding - kernel_size) // stride + 1
    wo = (w + 2 * padding - kernel_size) // stride + 1
    return rows.reshape(b * rows.size(1), -1), (b, ho, wo)


def fold_rows(rows, shape, out_channels):
    b, ho, wo = shape
    return rows.reshape(b, ho * wo, out_channels).transpose(1, 2).reshape(b, out_channels, ho, wo)


class _KANConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, layer):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.kanlayer = layer

    def forward(self, x):
        assert x.size(1) == self.in_channels
        rows, shape = unfold_rows(x, self.kernel_size, self.stride, self.padding)
        return fold_rows(self.kanlayer(rows), shape, self.out_channels)


class ChebyKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:40-68."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, degree=4):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding,
                         ChebyKANLayer(in_channels * kernel_size * kernel_size, out_channels, degree=degree))


class FastKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:71-100."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, FastKANLayer(in_channels * kernel_size * kernel_size, out_channels))


class GRAMKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:103-133."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, GRAMLayer(in_channels * kernel_size * kernel_size, out_channels))


class WavKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:136-165."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, wavelet_type="mexican_hat"):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding,
                         WavKANLayer(in_channels * kernel_size * kernel_size, out_channels, wavelet_type=wavelet_type))


class JacobiKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:168-198."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, degree=4):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding,
                         JacobiKANLayer(in_channels * kernel_size * kernel_size, out_channels, degree=degree))


class ReLUKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:201-231 (g = 5, k = 3)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding,
                         ReLUKANLayer(in_channels * kernel_size * kernel_size, 5, 3, out_channels))


class FasterKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:234-263."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, FasterKANLayer(in_channels * kernel_size * kernel_size, out_channels))


class RBFKANConv2d(_KANConvBase):
    """KANConv2Dlayers.py:266-293."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, RBFKANLayer(in_channels * kernel_size * kernel_size, out_channels))


VARIANTS = {"ChebyKANConv2d": ChebyKANConv2d, "FastKANConv2d": FastKANConv2d, "GRAMKANConv2d": GRAMKANConv2d, "WavKANConv2d": WavKANConv2d,
            "JacobiKANConv2d": JacobiKANConv2d, "ReLUKANConv2d": ReLUKANConv2d, "FasterKANConv2d": FasterKANConv2d, "RBFKANConv2d": RBFKANConv2d}
