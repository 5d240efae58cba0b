"""Graph-replay time of one sub-module's forward + backward at its bench shape (which blocks sit on the step's dependent chain
and what they cost there).  python tools/time_block.py bridge|lca3|iwp3|dec1|msf1|evim16|evim32|evim64 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd

torch.manual_seed(0)
dev = "cuda"
m = km_unet_amd.KM_UNetV3(num_classes=5).to(dev).train()
B = int(os.environ.get("TB_B", "8"))
specs = {
    "bridge": (lambda: m.bridge_attention, [(B, 64, 16, 16)]),
    "lca3": (lambda: m.lca3, [(B, 64, 16, 16)]),
    "enc1": (lambda: m.enc1, [(B, 16, 128, 128)]),
    "enc2": (lambda: m.enc2, [(B, 16, 64, 64)]),
    "enc3": (lambda: m.enc3, [(B, 32, 32, 32)]),
    "kan1": (lambda: m.enc1[0], [(B, 16, 128, 128)]),
    "vim16": (lambda: m.enc1[1], [(B, 16, 128, 128)]),
    "vim32": (lambda: m.enc2[1], [(B, 32, 64, 64)]),
    "vim64": (lambda: m.enc3[1], [(B, 64, 32, 32)]),
    "iwp1": (lambda: m.enc1[2], [(B, 16, 128, 128)]),
    "dec1": (lambda: m.dec1, [(B, 64, 16, 16)]),
    "dir16": (lambda: m.enc1[1].height_block, [(B, 16, 128, 128)]),
    "evim16": (lambda: m.enc1[1].height_block.vit_mamba, [(B, 16, 128, 128)]),
    "evim32": (lambda: m.enc2[1].height_block.vit_mamba, [(B, 32, 64, 64)]),
    "evim64": (lambda: m.enc3[1].height_block.vit_mamba, [(B, 64, 32, 32)]),
    "vimdec32": (lambda: m.dec2[2], [(B, 32, 64, 64)]),
    "iwp2": (lambda: m.enc2[2], [(B, 32, 64, 64)]),
    "lca1": (lambda: m.lca1, [(B, 16, 64, 64)]),
    "kan2": (lambda: m.enc2[0], [(B, 16, 64, 64)]),
    "kan3": (lambda: m.enc3[0], [(B, 32, 32, 32)]),
    "deckan": (lambda: m.dec1[1], [(B, 64, 32, 32)]),
    "up1": (lambda: m.dec1[0], [(B, 64, 16, 16)]),
    "up2": (lambda: m.dec2[0], [(B, 64, 32, 32)]),
    "up3": (lambda: m.dec3[0], [(B, 64, 64, 64)]),
    "msf1": (lambda: m.attention1[0], None),
    "dec2": (lambda: _Seq(m.dec2, lambda d, x: d[2](km_unet_amd.nn.conv3x3(d[0](x), d[1]))), [(B, 64, 32, 32)]),
    "dec3": (lambda: _Seq(m.dec3, lambda d, x: km_unet_amd.nn.conv3x3(d[2](km_unet_amd.nn.conv3x3(d[0](x), d[1])), d[3])), [(B, 64, 64, 64)]),
    "lca2": (lambda: m.lca2, [(B, 32, 32, 32)]),
}
class _Seq(torch.nn.Module):
    """a decoder stage run the way KM_UNetV3.forward runs it"""
    def __init__(self, d, fn):
        super().__init__()
        self.d, self.fn = d, fn
    def forward(self, x):
        return self.fn(self.d, x)


for name in sys.argv[1:]:
    mod, shapes = specs[name][0](), specs[name][1]
    if shapes is None:       # MultiScaleFusion takes a list of three features
        feats = [torch.randn(B, c, 32, 32, device=dev, requires_grad=True) for c in (16, 32, 32)]
        xs, inner = feats, mod
        mod = lambda *f: inner(list(f))
        mod.parameters = inner.parameters
    else:
        xs = [torch.randn(*s, device=dev, requires_grad=True) for s in shapes]
    params = [p for p in mod.parameters() if p.requires_grad]
    from km_unet_amd import ops
    defer = os.environ.get("TB_DEFER", "1") == "1"      # weight gradients queued and flushed at the end, as in the train step
    def step():
        y = mod(*xs)
        ops.WGRAD_OVERLAP = defer
        try:
            g = torch.autograd.grad(y.float().square().mean(), xs + params, allow_unused=True)
        finally:
            ops.WGRAD_OVERLAP = False
            ops.flush_wgrad_jobs(final=True)
        return g
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-8s fwd+bwd %8.1f us per replay" % (name, e0.elapsed_time(e1) / 30 * 1e3), flush=True)
