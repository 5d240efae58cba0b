import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import warnings; warnings.filterwarnings("ignore")
import torch
import km_unet_amd
from km_unet_amd import ops
from conftest import rel_err
from oracle.model import fill_parameters
from oracle import hsmssd as oh
B, C, H = 8, 16, 128
torch.manual_seed(C + H)
blk = fill_parameters(km_unet_amd.EfficientViMBlock(C, state_dim=64), 5).train().cuda()
ref = fill_parameters(oh.EfficientViMBlock(C, state_dim=64), 5).double().train()
x = torch.randn(B, C, H, H); gy = torch.randn(B, C, H, H)
# reference with retained intermediates
xr = x.double().requires_grad_(True)
t1r = ref.ffn.fc1.conv(xr); t1r.retain_grad()
hr = torch.relu(ref.ffn.fc1.norm(t1r)); hr.retain_grad()
t2r = ref.ffn.fc2.conv(hr); t2r.retain_grad()
a = torch.sigmoid(ref.alpha).view(4, -1, 1, 1)
yr = (1 - a[3]) * xr + a[3] * ref.ffn.fc2.norm(t2r)
yr.backward(gy.double())
for trial in range(2):
    xg = x.cuda().requires_grad_(True)
    t1 = blk.ffn.fc1.conv_only(xg); t1.retain_grad()
    h = ops.bn_blend(t1, None, blk.ffn.fc1.norm, None, 0, relu=True); h.retain_grad()
    t2 = blk.ffn.fc2.conv_only(h); t2.retain_grad()
    y = ops.bn_blend(t2, xg, blk.ffn.fc2.norm, blk.alpha, 3)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    print("trial", trial, "y=%.1e  d_t2=%.1e  d_h=%.1e  d_t1=%.1e  dx=%.1e" % (rel_err(y, yr), rel_err(t2.grad, t2r.grad), rel_err(h.grad, hr.grad), rel_err(t1.grad, t1r.grad), rel_err(xg.grad, xr.grad)))
    for name, a_, b_ in (("d_t2", t2.grad, t2r.grad), ("d_h", h.grad, hr.grad), ("d_t1", t1.grad, t1r.grad), ("dx", xg.grad, xr.grad)):
        d = (a_.cpu().double() - b_).abs() > 1e-4 * b_.abs().max()
        if d.any():
            idx = d.nonzero()
            print("   %s bad: %d of %d ; b in %s ; c in %s ; h range %d..%d ; w range %d..%d" % (name, int(d.sum()), d.numel(),
                  sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))[:20], int(idx[:, 2].min()), int(idx[:, 2].max()), int(idx[:, 3].min()), int(idx[:, 3].max())))
    for p in blk.parameters(): p.grad = None
