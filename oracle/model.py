"""Oracle: whole KM_UNetV3 graph, SH and LAPS variants (TEST INFRASTRUCTURE).

Follows KM_UNetV3_SH.py:21-517 (and KM_UNetV3_LAPS.py, which differs only in
having no DAGEM bridge and bilinear nn.Upsample instead of DySample).  Module
attribute names reproduce the reference state_dict keys exactly (920 keys for
SH num_classes=20) including the parameters the forward never touches
(branches.plain, attn, dt_proj -- SURVEY quirk 6).

Runs on CPU in fp32: the reference's @autocast() decorators are CUDA-only and
self-disable on CPU, which is the path this oracle restates.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .dagem import DAGEM
from .dysample import DySample
from .hsmssd import EfficientViMBlock
from .iwp import IntelligentWaveletPoolingModule
from .kan import KANConv2d


class DropPath(nn.Module):
    """timm 0.9.16 DropPath restated (third-party; eval => identity)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.drop_prob = p

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class StableHybridKANConv(nn.Module):
    """KM_UNetV3_SH.py:21-94: GroupNorm(4) -> residual 1x1 -> KANConv2d -> ReLU(id + kan)."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.branches = nn.ModuleDict({"plain": KANConv2d(cin, cout, kernel_size, padding=padding)})  # unused twin
        self.kanconv2d = nn.Sequential(KANConv2d(cin, cout, kernel_size, padding=padding))
        self.attn = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, 1, 1), nn.Softmax(dim=1))      # unused
        self.pre_norm = nn.GroupNorm(4, cin)
        self.post_act = nn.ReLU(inplace=True)
        self.residual = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()

    def forward(self, x):
        x = self.pre_norm(x)
        return self.post_act(self.residual(x) + self.kanconv2d(x))


class DirectionAttention(nn.Module):
    """KM_UNetV3_SH.py:215-263.  All three pooling modes reduce to the global
    mean over (H, W) (mean of row/column means)."""

    def __init__(self, dim, mode):
        super().__init__()
        self.mode = mode
        self.qkv = nn.Conv2d(dim, dim * 3, 1)
        self.conv = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.fc = nn.Sequential(nn.Linear(dim, dim // 4), nn.GELU(), nn.Linear(dim // 4, dim), nn.Sigmoid())

    def forward(self, x):
        b, c = x.shape[:2]
        if self.mode == "height":
            pooled = x.mean(dim=3).mean(dim=2)
        elif self.mode == "width":
            pooled = x.mean(dim=2).mean(dim=2)
        else:
            pooled = x.mean(dim=(2, 3))
        q, k, v = self.qkv(x).chunk(3, dim=1)
        return self.conv(torch.sigmoid(q * k) * v) * self.fc(pooled).view(b, c, 1, 1)


class DirectionViM(nn.Module):
    """KM_UNetV3_SH.py:154-212.  state_dim is hard-wired to 64 (quirk 4)."""

    def __init__(self, dim, mode="height", state_dim=64):
        super().__init__()
        self.dt_proj = nn.Linear(dim, state_dim)          # never used in forward
        self.vit_mamba = EfficientViMBlock(dim, mlp_ratio=4, ssd_expand=1, state_dim=64)
        if mode == "height":
            self.proj = nn.Conv2d(dim, dim, (3, 1), padding=(1, 0))
        elif mode == "width":
            self.proj = nn.Conv2d(dim, dim, (1, 3), padding=(0, 1))
        else:
            self.proj = nn.Conv2d(dim, dim, 1)
        self.attn = DirectionAttention(dim, mode)

    def forward(self, x):
        return self.attn(self.vit_mamba(self.proj(x)))


class TripleNorm(nn.Module):
    """KM_UNetV3_SH.py:266-284."""

    def __init__(self, dim):
        super().__init__()
        self.norm_h = nn.GroupNorm(1, dim)
        self.norm_w = nn.GroupNorm(1, dim)
        self.norm_c = nn.LayerNorm(dim)

    def forward(self, x):
        h = self.norm_h(x.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
        c = self.norm_c(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return (h + self.norm_w(x) + c) / 3


class EnhancedViMBlock(nn.Module):
    """KM_UNetV3_SH.py:97-151."""

    def __init__(self, dim, expansion=4, state_dim=64, drop_path=0.1):
        super().__init__()
        self.height_block = DirectionViM(dim, "height", state_dim)
        self.width_block = DirectionViM(dim, "width", state_dim)
        self.channel_block = DirectionViM(dim, "channel", state_dim)
        self.fusion_gate = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(dim * 3, dim // 4, 1), nn.GELU(),
                                         nn.Conv2d(dim // 4, 3, 1), nn.Softmax(dim=1))
        self.ffn = nn.Sequential(nn.Conv2d(dim, dim * expansion, 1), nn.GELU(), nn.Conv2d(dim * expansion, dim, 1))
        self.norm = TripleNorm(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0 else nn.Identity()

    def forward(self, x):
        hf, wf, cf = self.height_block(x), self.width_block(x), self.channel_block(x)
        g = self.fusion_gate(torch.cat([hf, wf, cf], dim=1))
        x = x + self.drop_path(g[:, 0:1] * hf + g[:, 1:2] * wf + g[:, 2:3] * cf)
        return x + self.drop_path(self.ffn(self.norm(x)))


class ChannelAttention(nn.Module):
    def __init__(self, channel, reduction=8):
        super().__init__()
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.SiLU(),
                                nn.Linear(channel // reduction, channel), nn.Sigmoid())

    def forward(self, x):
        b, c = x.shape[:2]
        return x * self.fc(self.gap(x).view(b, c)).view(b, c, 1, 1)


class MultiScaleFusion(nn.Module):
    """KM_UNetV3_SH.py:287-311."""

    def __init__(self, channels, reduction=4):
        super().__init__()
        co = channels[-1]
        self.blocks = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, co, s, padding=s // 2), nn.GroupNorm(1, co), nn.SiLU())
            for c, s in zip(channels, [3, 5, 7])])
        self.fusion = nn.Sequential(nn.Conv2d(co * 3, co, 1), nn.Conv2d(co, co, 3, padding=1),
                                    ChannelAttention(co, reduction))

    def forward(self, feats):
        return self.fusion(torch.cat([blk(f) for blk, f in zip(self.blocks, feats)], dim=1))


class LocalContrastAttention(nn.Module):
    """KM_UNetV3_SH.py:336-368."""

    def __init__(self, in_channels, reduction_ratio=4):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.reduction_ratio = reduction_ratio
        self.fc = nn.Sequential(nn.Linear(in_channels // reduction_ratio, 64), nn.ReLU(),
                                nn.Linear(64, in_channels), nn.Sigmoid())

    def forward(self, x):
        avg = x.mean(dim=(2, 3))
        g = self.fc(avg.view(avg.shape[0], -1, self.reduction_ratio).mean(-1))[:, :, None, None]
        return x * (1 - g) + g


class KM_UNetV3(nn.Module):
    """variant='SH' (KM_UNetV3_SH.py:371-517) or 'LAPS' (KM_UNetV3_LAPS.py:366-...)."""

    def __init__(self, num_classes=3, embed_dims=(16, 32, 64), variant="SH"):
        super().__init__()
        e0, e1, e2 = embed_dims
        sh = variant == "SH"
        up = (lambda: DySample(e2, scale=2, style="lp")) if sh else \
             (lambda: nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))
        self.conv_f = nn.Conv2d(5, 16, 3, padding=1)
        self.lca1, self.lca2, self.lca3 = (LocalContrastAttention(e) for e in (e0, e1, e2))
        self.enc1 = nn.Sequential(StableHybridKANConv(16, e0), EnhancedViMBlock(e0, state_dim=16),
                                  IntelligentWaveletPoolingModule(e0))
        self.enc2 = nn.Sequential(StableHybridKANConv(e0, e1), EnhancedViMBlock(e1, state_dim=16),
                                  IntelligentWaveletPoolingModule(e1))
        self.enc3 = nn.Sequential(StableHybridKANConv(e1, e2), EnhancedViMBlock(e2, state_dim=16),
                                  IntelligentWaveletPoolingModule(e2))
        if sh:
            self.bridge_attention = DAGEM(sync_bn=False, input_channels=e2)
        self.dec1 = nn.Sequential(up(), StableHybridKANConv(e2, e1))
        self.attention1 = nn.Sequential(MultiScaleFusion([e0, e1, e1]))
        self.attention2 = nn.Sequential(MultiScaleFusion([e0, e1, e1]))
        self.dec2 = nn.Sequential(up(), nn.Conv2d(e1 * 2, e1, 3, padding=1), EnhancedViMBlock(e1, state_dim=16))
        self.dec3 = nn.Sequential(up(), nn.Conv2d(e1 * 2, e0, 3, padding=1), EnhancedViMBlock(e0),
                                  nn.Conv2d(e0, num_classes, 3, padding=1))
        self.output_norm = nn.GroupNorm(1, num_classes)
        self.activation = nn.Sigmoid()
        self.variant = variant

    def forward(self, x):
        x = self.conv_f(x)
        e1 = self.lca1(self.enc1(x))
        e2 = self.lca2(self.enc2(e1))
        e3 = self.lca3(self.enc3(e2))
        d1 = self.dec1(self.bridge_attention(e3) if self.variant == "SH" else e3)
        rs = lambda t, ref: F.interpolate(t, size=ref.shape[2:], mode="bilinear", align_corners=True)
        # the third pyramid input is e2 again, not e3 (KM_UNetV3_SH.py:495,509; quirk 8)
        d1 = torch.cat([d1, self.attention1([rs(e1, d1), rs(e2, d1), rs(e2, d1)])], dim=1)
        d2 = self.dec2(d1)
        d2 = torch.cat([d2, self.attention2([rs(e1, d2), rs(e2, d2), rs(e2, d2)])], dim=1)
        return self.activation(self.output_norm(self.dec3(d2)))


def fill_parameters(model, seed=0):
    """Deterministic, reference-independent weights for parity fixtures.

    Every floating-point tensor of the state_dict except the KAN ``grid`` and
    DySample ``init_pos`` buffers is overwritten from a CPU generator seeded by
    (seed, position in sorted key order); BatchNorm running_var stays positive.
    The same function runs on the reference model in make_golden.py and on the
    oracle / product models in the tests, so no weights need to be stored.
    """
    sd = model.state_dict()
    with torch.no_grad():
        for i, key in enumerate(sorted(sd)):
            t = sd[key]
            if not t.is_floating_point() or key.endswith(".grid") or key.endswith("init_pos"):
                continue
            g = torch.Generator().manual_seed(seed * 100003 + i)
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            if key.endswith("running_var"):
                v = 0.5 + r.abs()
            elif key.endswith("running_mean"):
                v = 0.1 * r
            elif t.ndim <= 1 or key.endswith("norm.weight") or key.endswith(".alpha"):
                v = (1.0 if ("norm" in key and key.endswith("weight")) else 0.0) + 0.2 * r
                if key.endswith(".D"):
                    v = 1.0 + 0.2 * r
            else:
                fan_in = max(1, t[0].numel())
                v = r * (1.0 / fan_in) ** 0.5
            t.copy_(v.to(t.dtype))
    return model
