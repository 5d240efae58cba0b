#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UPSTREAM REFERENCE on CPU (fp32).

Container-only: needs /root/reference mounted (it never travels to the GPU
box).  Re-run with ``python tests/golden/make_golden.py`` from the repo root.
Fixtures hold data only -- seeded inputs, explicit parameters and the
reference's outputs / gradients -- never reference source.

What each fixture pins (reference file:line):
  k1_*.npz     convKAN/KANConv2Dlayers.py:15-37 + KANlayers.py:577-660
  kan_reg.npz  KANlayers.py:713-731 (regularization_loss)
  k2_*.npz     vim_block_init/efficient_vim_init.py:33-61 (+ LayerNorm1D vim_utils_init.py:50-59)
  evim_*.npz   vim_block_init/efficient_vim_init.py:81-97 (EfficientViMBlock, train + eval BN)
  k3_*.npz     DySample_md.py:49-68 (+ integer gather indices from the oracle's
               index-explicit form after it reproduced F.grid_sample's output)
  iwp_*.npz    WPL/iwp.py:116-132
  dagem_plain_*.npz  DAGEM_md.py:56-111 with the deformable conv replaced by a plain conv on both sides
  kanvar_*.npz convKAN/KANConv2Dlayers.py:40-293 (the eight alternative KAN convolutions: output + input gradient for explicit parameters)
  kan_update_grid.npz  KANlayers.py:662-709 (update_grid: knots and spline coefficients after the re-fit, layer output before / after)
  model_*.npz  KM_UNetV3_SH.py:465-517 / KM_UNetV3_LAPS.py (DAGEM's deform-conv
               = oracle restatement of torchvision => that sub-block is unpinned)
  manifest_*.txt  state_dict key / shape / dtype lists
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import ref_loader  # noqa: E402
from oracle.dysample import dysample_lp_indices  # noqa: E402
from oracle.model import fill_parameters  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


def set_params(mod, seed, scale_fn=None):
    with torch.no_grad():
        for i, (k, p) in enumerate(sorted(mod.named_parameters())):
            s = scale_fn(k, p) if scale_fn else (1.0 / max(1, p[0].numel()) ** 0.5 if p.ndim > 1 else 0.3)
            p.copy_(rnd(seed * 1000 + i, *p.shape, scale=s))


# ---------------------------------------------------------------- K1
def gen_k1(ref):
    cases = {
        # name: (B, Cin, Cout, H, W, x-scale)
        "k1_s0": (2, 16, 16, 12, 12, 1.0),
        "k1_s1": (1, 16, 32, 9, 7, 0.6),          # ragged, non-square, Cin != Cout
        "k1_oos": (1, 4, 16, 8, 8, 3.0),           # most |x| outside the spline support [-2.2, 2.2)
        "k1_border": (1, 4, 4, 2, 2, 0.5),         # every tap set touches the zero padding
        "k1_wide": (1, 64, 32, 6, 6, 1.0),         # dec1 shape 64 -> 32
    }
    for si, (name, (b, ci, co, h, w, sc)) in enumerate(cases.items()):
        m = ref.kanconv.KANConv2d(ci, co, 3, padding=1)
        set_params(m, 10 + si, lambda k, p: 0.3 if "scaler" in k else (0.2 if "spline" in k else 0.15))
        x = rnd(100 + si, b, ci, h, w, scale=sc).requires_grad_(True)
        if name == "k1_oos":
            with torch.no_grad():   # exact knot hits exercise the half-open intervals
                x.view(-1)[:6] = torch.tensor([-2.2, 2.2, -1.0, 1.0, 0.2, 2.1999998])
        y = m(x)
        gy = rnd(200 + si, *y.shape)
        y.backward(gy)
        k = m.kanlayer
        save(name, x=x, gy=gy, grid=k.grid, base_weight=k.base_weight, spline_weight=k.spline_weight,
             spline_scaler=k.spline_scaler, y=y, dx=x.grad, d_base_weight=k.base_weight.grad,
             d_spline_weight=k.spline_weight.grad, d_spline_scaler=k.spline_scaler.grad)


def gen_kan_reg(ref):
    """KANLinear.regularization_loss (KANlayers.py:713-731) of the reference for explicit spline weights."""
    m = ref.kanconv.KANConv2d(4, 8, 3, padding=1).kanlayer
    set_params(m, 77, lambda k, p: 0.2)
    save("kan_reg", spline_weight=m.spline_weight, r11=m.regularization_loss(1.0, 1.0), r03_2=m.regularization_loss(0.3, 2.0))


def gen_kan_variants(ref):
    """The eight alternative KAN convolutions (KANConv2Dlayers.py:40-293) on one small input each, parameters set explicitly."""
    kw = {"ChebyKANConv2d": {"degree": 4}, "WavKANConv2d": {"wavelet_type": "mexican_hat"}, "JacobiKANConv2d": {"degree": 4}}
    for si, name in enumerate(("ChebyKANConv2d", "FastKANConv2d", "GRAMKANConv2d", "WavKANConv2d", "JacobiKANConv2d", "ReLUKANConv2d",
                               "FasterKANConv2d", "RBFKANConv2d")):
        torch.manual_seed(900 + si)
        m = getattr(ref.kanconv, name)(3, 5, 3, padding=1, **kw.get(name, {}))
        m.eval()                                    # WavKAN's BatchNorm1d: running statistics (deterministic)
        with torch.no_grad():                       # move every trainable tensor off its initial value, keep scales positive
            for i, (k, p_) in enumerate(sorted(m.named_parameters())):
                if not p_.requires_grad:
                    continue
                p_.add_(rnd(900 * 100 + si * 50 + i, *p_.shape, scale=0.1 if p_.ndim > 0 else 0.0))
            for k, b_ in m.named_buffers():
                if k.endswith("running_var"):
                    b_.copy_(torch.rand(b_.shape, generator=torch.Generator().manual_seed(si)) + 0.5)
                if k.endswith("running_mean"):
                    b_.copy_(rnd(si + 5, *b_.shape, scale=0.2))
        x = rnd(950 + si, 2, 3, 6, 5, scale=0.8).requires_grad_(True)
        y = m(x)
        gy = rnd(970 + si, *y.shape)
        y.backward(gy)
        sd = {"sd__" + k.replace(".", "__"): v for k, v in m.state_dict().items()}
        save("kanvar_" + name, x=x, gy=gy, y=y, dx=x.grad, **sd)
    # the other wavelets of WavKANLayer (KANlayers.py:262-300) share everything but psi: forward only
    for kind in ("morlet", "dog", "meyer", "shannon"):
        torch.manual_seed(990)
        m = ref.kanconv.WavKANConv2d(2, 3, 3, padding=1, wavelet_type=kind).eval()
        x = rnd(991, 1, 2, 4, 4, scale=0.8)
        with torch.no_grad():
            y = m(x)
        save("kanvar_wav_" + kind, x=x, y=y, **{"sd__" + k.replace(".", "__"): v for k, v in m.state_dict().items()})


def gen_update_grid(ref):
    """KANLinear.update_grid (KANlayers.py:662-709) on explicit parameters and rows."""
    torch.manual_seed(5)
    m = ref.kanlayers.KANLinear(6, 4)
    set_params(m, 81, lambda k, p: 0.3 if "scaler" in k else 0.2)
    x = rnd(82, 40, 6, scale=0.7)
    before = dict(grid=m.grid.clone(), spline_weight=m.spline_weight.detach().clone(), base_weight=m.base_weight.detach().clone(),
                  spline_scaler=m.spline_scaler.detach().clone())
    y0 = m(x)
    m.update_grid(x)
    y1 = m(x)
    save("kan_update_grid", x=x, y_before=y0, y_after=y1, grid_after=m.grid, spline_weight_after=m.spline_weight, **{k + "_before": v for k, v in before.items()})


# ---------------------------------------------------------------- K2
def gen_k2(ref):
    cases = {"k2_c16": (2, 16, 8), "k2_c64": (1, 64, 4), "k2_c32": (2, 32, 6)}
    for si, (name, (b, c, hh)) in enumerate(cases.items()):
        m = ref.vim.HSMSSD(d_model=c, state_dim=64)
        ln = ref.vim_utils.LayerNorm1D(c)
        set_params(m, 30 + si, lambda k, p: 0.4 if k in ("A", "D") or "dw" in k else 1.0 / p[0].numel() ** 0.5)
        with torch.no_grad():
            m.D.add_(1.0)
            m.A.copy_(torch.rand(64) * 15 + 1)
            ln.weight.copy_(1 + rnd(40 + si, 1, c, 1, scale=0.2))
            ln.bias.copy_(rnd(41 + si, 1, c, 1, scale=0.2))
        x0 = rnd(300 + si, b, c, hh * hh).requires_grad_(True)
        xn = ln(x0)
        xn.retain_grad()
        y, h = m(xn)
        gy, gh = rnd(310 + si, *y.shape), rnd(320 + si, *h.shape, scale=0.1)
        (y * gy).sum().add((h * gh).sum()).backward()
        save(name, x0=x0, ln_weight=ln.weight, ln_bias=ln.bias, xn=xn, gy=gy, gh=gh,
             w_bcdt=m.BCdt_proj.conv.weight, w_dw=m.dw.conv.weight, w_hz=m.hz_proj.conv.weight,
             w_out=m.out_proj.conv.weight, A=m.A, D=m.D, y=y, h=h,
             d_xn=xn.grad, d_x0=x0.grad, d_ln_weight=ln.weight.grad, d_ln_bias=ln.bias.grad,
             d_w_bcdt=m.BCdt_proj.conv.weight.grad, d_w_dw=m.dw.conv.weight.grad,
             d_w_hz=m.hz_proj.conv.weight.grad, d_w_out=m.out_proj.conv.weight.grad, d_A=m.A.grad, d_D=m.D.grad)


def gen_evim(ref):
    for si, (name, train) in enumerate({"evim_eval": False, "evim_train": True}.items()):
        m = ref.vim.EfficientViMBlock(dim=16, state_dim=64)
        fill_parameters(m, 7 + si)
        m.train(train)
        x = rnd(400 + si, 2, 16, 8, 8).requires_grad_(True)
        y = m(x)
        gy = rnd(410 + si, *y.shape)
        y.backward(gy)
        gsum = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        save(name, x=x, gy=gy, y=y, dx=x.grad,
             grad_keys=np.array(sorted(gsum)), grad_sums=np.array([gsum[k].double().sum().item() for k in sorted(gsum)]),
             grad_abs=np.array([gsum[k].double().abs().sum().item() for k in sorted(gsum)]))


# ---------------------------------------------------------------- K3
def gen_k3(ref):
    cases = {"k3_default": (1, 64, 6, 6, None), "k3_large": (1, 64, 6, 6, 0.5), "k3_b2": (2, 64, 5, 5, 0.2),
             "k3_16": (1, 64, 16, 16, 0.3)}
    for si, (name, (b, c, h, w, wstd)) in enumerate(cases.items()):
        m = ref.dysample.DySample(c, scale=2, style="lp")
        if wstd is not None:
            with torch.no_grad():
                m.offset.weight.copy_(rnd(50 + si, *m.offset.weight.shape, scale=wstd))
                m.offset.bias.copy_(rnd(51 + si, *m.offset.bias.shape, scale=wstd))
        else:
            with torch.no_grad():
                m.offset.weight.copy_(rnd(50 + si, *m.offset.weight.shape, scale=0.001))
        x = rnd(500 + si, b, c, h, w).requires_grad_(True)
        y = m(x)
        gy = rnd(510 + si, *y.shape)
        y.backward(gy)
        with torch.no_grad():
            yo, ix0, iy0 = dysample_lp_indices(x, m.offset.weight, m.offset.bias, m.init_pos)
        err = (yo - y).abs().max().item()
        assert err < 2e-5, (name, err)      # index-explicit oracle reproduces F.grid_sample
        save(name, x=x, gy=gy, w_off=m.offset.weight, b_off=m.offset.bias, init_pos=m.init_pos, y=y,
             ix0=ix0, iy0=iy0, dx=x.grad, d_w_off=m.offset.weight.grad, d_b_off=m.offset.bias.grad)


def gen_iwp(ref):
    m = ref.iwp.IntelligentWaveletPoolingModule(16)
    fill_parameters(m, 3)
    x = rnd(600, 2, 16, 8, 8).requires_grad_(True)
    y = m(x)
    gy = rnd(601, *y.shape)
    y.backward(gy)
    save("iwp_c16", x=x, gy=gy, y=y, dx=x.grad)


def gen_dagem(ref):
    """DAGEM_md.py:56-111 with the deformable conv replaced by oracle.dagem.plain_conv_stand_in on the REFERENCE module:
    pins the block's own arithmetic (edges, MLPs, BatchNorms, final layer) independent of torchvision."""
    from oracle.dagem import plain_conv_stand_in
    for si, (name, train) in enumerate({"dagem_plain_eval": False, "dagem_plain_train": True}.items()):
        m = plain_conv_stand_in(ref.dagem.DAGEM(sync_bn=False, input_channels=64))
        fill_parameters(m, 11 + si)
        m.train(train)
        x = rnd(800 + si, 2, 64, 8, 8, scale=0.7).requires_grad_(True)
        y = m(x)
        gy = rnd(810 + si, *y.shape)
        y.backward(gy)
        g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        keys = sorted(g)
        save(name, x=x, gy=gy, y=y, dx=x.grad, grad_keys=np.array(keys),
             **{"g__" + k.replace(".", "__"): g[k] for k in keys})


# ---------------------------------------------------------------- whole model
def gen_model(ref):
    for name, mod, variant, nc, train, b, hw in [
        ("model_sh_eval", ref.sh, "SH", 5, False, 1, 32),
        ("model_sh_train", ref.sh, "SH", 5, True, 2, 64),   # 64x64: >= 128 samples per BatchNorm channel at every level
        ("model_laps_eval", ref.laps, "LAPS", 3, False, 1, 32),
    ]:
        torch.manual_seed(0)
        m = mod.KM_UNetV3(num_classes=nc)
        fill_parameters(m, 1)
        m.train(train)
        for sub in m.modules():                 # DropPath is third-party RNG: disabled for parity
            if hasattr(sub, "drop_prob"):
                sub.drop_prob = 0.0
        x = torch.rand(b, 5, hw, hw, generator=torch.Generator().manual_seed(700)).requires_grad_(True)
        tgt = torch.rand(b, nc, hw, hw, generator=torch.Generator().manual_seed(701))
        y = m(x)
        loss = torch.nn.functional.mse_loss(y, tgt)
        loss.backward()
        g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        keys = sorted(g)
        full = {"g__" + k.replace(".", "__"): g[k] for k in keys
                if ("enc1.0.kanconv2d" in k or "enc1.1.height_block.vit_mamba.mixer" in k or k.startswith("dec1.0.offset")
                    or k.startswith("conv_f"))}
        save(name, x=x, target=tgt, y=y, loss=loss, dx=x.grad, grad_keys=np.array(keys),
             grad_sums=np.array([g[k].double().sum().item() for k in keys]),
             grad_abs=np.array([g[k].double().abs().sum().item() for k in keys]),
             n_no_grad=np.array(sum(1 for _, p in m.named_parameters() if p.grad is None)), **full)
    for fname, mod, nc in [("manifest_sh_nc20.txt", ref.sh, 20), ("manifest_laps_nc3.txt", ref.laps, 3)]:
        m = mod.KM_UNetV3(num_classes=nc)
        with open(os.path.join(OUT, fname), "w") as f:
            for k, v in m.state_dict().items():
                f.write("%s %s %s\n" % (k, "x".join(map(str, v.shape)) or "scalar", str(v.dtype).replace("torch.", "")))
        print("wrote", fname)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = ref_loader.load()
    if "--only-variants" in sys.argv:
        gen_kan_variants(ref)
        gen_update_grid(ref)
        raise SystemExit(0)
    gen_k1(ref)
    gen_kan_reg(ref)
    gen_kan_variants(ref)
    gen_update_grid(ref)
    gen_k2(ref)
    gen_evim(ref)
    gen_k3(ref)
    gen_iwp(ref)
    gen_dagem(ref)
    gen_model(ref)
