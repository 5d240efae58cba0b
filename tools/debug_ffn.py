"""Debug probe (GPU): FFN = conv1x1 -> BN+ReLU -> conv1x1 -> BN -> blend at model-like shapes vs fp64 CPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import warnings; warnings.filterwarnings("ignore")
import copy, torch
import km_unet_amd
from km_unet_amd import nn as knn, ops
from conftest import rel_err
from oracle.model import fill_parameters
from oracle import hsmssd as oh

for (B, C, H) in [(2, 16, 8), (2, 16, 64), (2, 32, 32), (2, 64, 16), (8, 16, 128)]:
    torch.manual_seed(C + H)
    blk = fill_parameters(km_unet_amd.EfficientViMBlock(C, state_dim=64), 5).train()
    ref = fill_parameters(oh.EfficientViMBlock(C, state_dim=64), 5).double().train()
    x = torch.randn(B, C, H, H)
    gy = torch.randn(B, C, H, H)
    xr = x.double().requires_grad_(True)
    # reference: FFN blend only (x <- lerp(x, ffn(x), a3)) in fp64
    a = torch.sigmoid(ref.alpha).view(4, -1, 1, 1)
    yr = (1 - a[3]) * xr + a[3] * ref.ffn(xr)
    yr.backward(gy.double())
    gm = blk.cuda()
    xg = x.cuda().requires_grad_(True)
    h = gm.ffn.fc1(xg)
    yg = ops.bn_blend(gm.ffn.fc2.conv_only(h), xg, gm.ffn.fc2.norm, gm.alpha, 3)
    yg.backward(gy.cuda())
    errs = {"y": rel_err(yg, yr), "dx": rel_err(xg.grad, xr.grad),
            "dW1": rel_err(gm.ffn.fc1.conv.weight.grad, ref.ffn.fc1.conv.weight.grad),
            "dW2": rel_err(gm.ffn.fc2.conv.weight.grad, ref.ffn.fc2.conv.weight.grad),
            "dg1": rel_err(gm.ffn.fc1.norm.weight.grad, ref.ffn.fc1.norm.weight.grad),
            "dg2": rel_err(gm.ffn.fc2.norm.weight.grad, ref.ffn.fc2.norm.weight.grad),
            "dalpha": rel_err(gm.alpha.grad[3], ref.alpha.grad[3])}
    print((B, C, H), "  ".join("%s=%.1e" % kv for kv in errs.items()))
    # whole block
    blk2 = fill_parameters(km_unet_amd.EfficientViMBlock(C, state_dim=64), 5).train().cuda()
    ref2 = fill_parameters(oh.EfficientViMBlock(C, state_dim=64), 5).double().train()
    xr2 = x.double().requires_grad_(True); ref2(xr2).backward(gy.double())
    xg2 = x.cuda().requires_grad_(True); blk2(xg2).backward(gy.cuda())
    worst = max((rel_err(p.grad, dict(ref2.named_parameters())[k].grad), k) for k, p in blk2.named_parameters() if p.grad is not None and not k.endswith(".A") and p.numel() > 1)
    print("      whole block: dx=%.1e worst=%s" % (rel_err(xg2.grad, xr2.grad), worst))
