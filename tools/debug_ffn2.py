import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import warnings; warnings.filterwarnings("ignore")
import copy, torch, torch.nn as nn
import km_unet_amd
from km_unet_amd import ops
from conftest import rel_err
torch.manual_seed(0)
B, Ci, Co, H = 8, 16, 64, 128
# (a) bmm with expanded weight, fwd + bwd, vs CPU fp64
x = torch.randn(B, Ci, H * H); w = torch.randn(Co, Ci) / 4; gy = torch.randn(B, Co, H * H)
xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
yr = torch.matmul(wr, xr); yr.backward(gy.double())
xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
yg = torch.bmm(wg.view(1, Co, Ci).expand(B, Co, Ci), xg); yg.backward(gy.cuda())
print("bmm expand:  y=%.1e dx=%.1e dw=%.1e" % (rel_err(yg, yr), rel_err(xg.grad, xr.grad), rel_err(wg.grad, wr.grad)))
# reverse shape (fc2): Co=16 from Ci=64
x2 = torch.randn(B, Co, H * H); w2 = torch.randn(Ci, Co) / 8; gy2 = torch.randn(B, Ci, H * H)
xr2, wr2 = x2.double().requires_grad_(True), w2.double().requires_grad_(True)
yr2 = torch.matmul(wr2, xr2); yr2.backward(gy2.double())
xg2, wg2 = x2.cuda().requires_grad_(True), w2.cuda().requires_grad_(True)
yg2 = torch.bmm(wg2.view(1, Ci, Co).expand(B, Ci, Co), xg2); yg2.backward(gy2.cuda())
print("bmm expand2: y=%.1e dx=%.1e dw=%.1e" % (rel_err(yg2, yr2), rel_err(xg2.grad, xr2.grad), rel_err(wg2.grad, wr2.grad)))
d = (xg2.grad.cpu().double() - xr2.grad).abs()
print("   dx2 bad elements (>1e-4):", int((d > 1e-4 * xr2.grad.abs().max()).sum()), "of", d.numel())
# (b) BN+ReLU (no blend) at [8,64,128,128]
t = torch.randn(B, Co, H, H) * 1.5 + 0.3; g2 = torch.randn(B, Co, H, H)
m = nn.BatchNorm2d(Co).double().train(); tr = t.double().requires_grad_(True)
torch.relu(m(tr)).backward(g2.double())
md = nn.BatchNorm2d(Co).cuda().train(); tg = t.cuda().requires_grad_(True)
ops.bn_blend(tg, None, md, None, 0, relu=True).backward(g2.cuda())
print("bn+relu big: dt=%.1e dgamma=%.1e" % (rel_err(tg.grad, tr.grad), rel_err(md.weight.grad, m.weight.grad)))
