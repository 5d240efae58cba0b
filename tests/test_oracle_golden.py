"""CPU: the oracle restatements reproduce the golden vectors the reference
produced (tests/golden/make_golden.py).  Tolerance: 1e-5 relative to the
tensor's max magnitude (both sides are fp32 CPU; differences are rounding
order only).  This is what "pins" the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import dysample as od
from oracle import hsmssd as oh
from oracle import kan as ok
from oracle.iwp import IntelligentWaveletPoolingModule
from oracle.model import KM_UNetV3, fill_parameters

TOL = 1e-5


@pytest.mark.parametrize("name", ["k1_s0", "k1_s1", "k1_oos", "k1_border", "k1_wide"])
def test_k1_kanconv2d(name):
    g = load_golden(name)
    x = g["x"].clone().requires_grad_(True)
    p = [g[k].clone().requires_grad_(True) for k in ("base_weight", "spline_weight", "spline_scaler")]
    y = ok.kan_conv2d(x, g["grid"], *p)
    assert rel_err(y, g["y"]) < TOL
    y.backward(g["gy"])
    assert rel_err(x.grad, g["dx"]) < TOL
    for t, k in zip(p, ("d_base_weight", "d_spline_weight", "d_spline_scaler")):
        assert rel_err(t.grad, g[k]) < TOL, k


@pytest.mark.parametrize("name", ["k2_c16", "k2_c32", "k2_c64"])
def test_k2_hsmssd(name):
    g = load_golden(name)
    x0 = g["x0"].clone().requires_grad_(True)
    names = ("w_bcdt", "w_dw", "w_hz", "w_out", "A", "D")
    p = {k: g[k].clone().requires_grad_(True) for k in names + ("ln_weight", "ln_bias")}
    xn = oh.layernorm1d(x0, p["ln_weight"], p["ln_bias"])
    assert rel_err(xn, g["xn"]) < TOL
    y, h = oh.hsmssd(xn, *[p[k] for k in names], state_dim=64)
    assert rel_err(y, g["y"]) < TOL and rel_err(h, g["h"]) < TOL
    ((y * g["gy"]).sum() + (h * g["gh"]).sum()).backward()
    assert rel_err(x0.grad, g["d_x0"]) < 5e-5
    for k in names[:4] + ("D", "ln_weight", "ln_bias"):
        assert rel_err(p[k].grad, g["d_" + k]) < 5e-5, k
    # A is a no-op parameter (softmax is shift invariant): its gradient is rounding noise
    assert g["d_A"].abs().max() < 1e-4 * max(1.0, g["d_w_bcdt"].abs().max().item())


@pytest.mark.parametrize("name,train", [("evim_eval", False), ("evim_train", True)])
def test_evim_block(name, train):
    g = load_golden(name)
    m = oh.EfficientViMBlock(16, state_dim=64)
    fill_parameters(m, 7 + int(train))
    m.train(train)
    x = g["x"].clone().requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["y"]) < TOL
    y.backward(g["gy"])
    assert rel_err(x.grad, g["dx"]) < 5e-5
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads) == list(g["grad_keys"])
    for k, s, a in zip(g["grad_keys"], g["grad_sums"], g["grad_abs"]):
        assert abs(grads[k].double().abs().sum().item() - a) <= 1e-4 * max(a, 1e-6) + 1e-7, k


@pytest.mark.parametrize("name,train", [("dagem_plain_eval", False), ("dagem_plain_train", True)])
def test_dagem_block_with_plain_conv_stand_in(name, train):
    """DAGEM_md.py:56-111 minus its one third-party operator: the reference's DAGEM and the oracle's, both with the deformable
    conv replaced by oracle.dagem.plain_conv_stand_in -- pins the roll-edge products, the four Linear+BatchNorm1d MLPs and
    the final aggregation layer (eval and batch-statistics mode)."""
    from oracle.dagem import DAGEM, plain_conv_stand_in
    g = load_golden(name)
    m = plain_conv_stand_in(DAGEM(sync_bn=False, input_channels=64))
    fill_parameters(m, 11 + int(train))
    m.train(train)
    x = g["x"].clone().requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["y"]) < TOL
    y.backward(g["gy"])
    assert rel_err(x.grad, g["dx"]) < 5e-5
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads) == list(g["grad_keys"])
    gmax = max(g["g__" + k.replace(".", "__")].abs().max().item() for k in g["grad_keys"])
    for k in g["grad_keys"]:       # biases in front of a batch-statistics BatchNorm: exactly-zero gradient, noise in the fixture
        ref = g["g__" + k.replace(".", "__")]
        assert (grads[k].double() - ref.double()).abs().max().item() / max(ref.abs().max().item(), 1e-4 * gmax) < 1e-4, k


@pytest.mark.parametrize("name", ["k3_default", "k3_large", "k3_b2", "k3_16"])
def test_k3_dysample(name):
    g = load_golden(name)
    x = g["x"].clone().requires_grad_(True)
    w = g["w_off"].clone().requires_grad_(True)
    b = g["b_off"].clone().requires_grad_(True)
    y = od.dysample_lp(x, w, b, g["init_pos"])
    assert rel_err(y, g["y"]) < TOL
    y.backward(g["gy"])
    assert rel_err(x.grad, g["dx"]) < TOL and rel_err(w.grad, g["d_w_off"]) < 5e-5 and rel_err(b.grad, g["d_b_off"]) < 5e-5
    yi, ix0, iy0 = od.dysample_lp_indices(g["x"], g["w_off"], g["b_off"], g["init_pos"])
    assert torch.equal(ix0, g["ix0"]) and torch.equal(iy0, g["iy0"])        # integer work: bit exact
    assert rel_err(yi, g["y"]) < 2e-5
    assert torch.equal(od.init_pos(), g["init_pos"])


def test_iwp():
    g = load_golden("iwp_c16")
    m = fill_parameters(IntelligentWaveletPoolingModule(16), 3)
    x = g["x"].clone().requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["y"]) < TOL
    y.backward(g["gy"])
    assert rel_err(x.grad, g["dx"]) < TOL


@pytest.mark.parametrize("name,variant,nc,train", [("model_sh_eval", "SH", 5, False), ("model_sh_train", "SH", 5, True),
                                                    ("model_laps_eval", "LAPS", 3, False)])
def test_whole_model(name, variant, nc, train):
    g = load_golden(name)
    m = fill_parameters(KM_UNetV3(num_classes=nc, variant=variant), 1)
    m.train(train)
    for sub in m.modules():
        if hasattr(sub, "drop_prob"):
            sub.drop_prob = 0.0
    x = g["x"].clone().requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["y"]) < 2e-5
    loss = torch.nn.functional.mse_loss(y, g["target"])
    assert abs(loss.item() - g["loss"].item()) < 1e-6
    loss.backward()
    assert rel_err(x.grad, g["dx"]) < 1e-3          # deep-net fp32 reordering noise; north_star tolerance
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads) == list(g["grad_keys"])
    assert sum(1 for _, p in m.named_parameters() if p.grad is None) == int(g["n_no_grad"])
    for k in g:
        if k.startswith("g__") and not k.endswith("__A"):
            assert rel_err(grads[k[3:].replace("__", ".")], g[k]) < 1e-3, k


@pytest.mark.parametrize("fname,variant,nc", [("manifest_sh_nc20.txt", "SH", 20), ("manifest_laps_nc3.txt", "LAPS", 3)])
def test_state_dict_manifest(fname, variant, nc):
    import os
    from conftest import GOLDEN
    want = [l.split() for l in open(os.path.join(GOLDEN, fname)).read().splitlines()]
    sd = KM_UNetV3(num_classes=nc, variant=variant).state_dict()
    got = [[k, "x".join(map(str, v.shape)) or "scalar", str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    assert sorted(map(tuple, got)) == sorted(map(tuple, want))
    if variant == "SH":
        assert len(got) == 920


def test_kan_regularization_loss_golden():
    """oracle.kan.regularization_loss and the product's KANLinear.regularization_loss against the reference's values."""
    import km_unet_amd
    from oracle import kan as okan
    g = load_golden("kan_reg")
    sw = g["spline_weight"]
    assert torch.allclose(okan.regularization_loss(sw, 1.0, 1.0), g["r11"], rtol=1e-6)
    assert torch.allclose(okan.regularization_loss(sw, 0.3, 2.0), g["r03_2"], rtol=1e-6)
    m = km_unet_amd.KANLinear(sw.shape[1], sw.shape[0])
    with torch.no_grad():
        m.spline_weight.copy_(sw)
    assert torch.allclose(m.regularization_loss(1.0, 1.0), g["r11"], rtol=1e-6)
    assert torch.allclose(m.regularization_loss(0.3, 2.0), g["r03_2"], rtol=1e-6)
