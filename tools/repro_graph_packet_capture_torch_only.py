"""Does the hipGraph replay corruption (loss 0.43 -> 216 at the third replay with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1) need
any kernel of this repository?  This script uses ONLY stock torch ops: a small conv net with per-sample stochastic depth
(bernoulli_ on the captured Philox stream), the tensor-op HybridLoss (aminmax, reflect pad, cat, depthwise conv, clamp,
means) and a capturable fused AdamW, captured into one torch.cuda.CUDAGraph and replayed six times.

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/repro_graph_packet_capture_torch_only.py [--blocks 24] [--droppath 1]
"""
import argparse
import os
import torch
import torch.nn as nn
import torch.nn.functional as F


class Block(nn.Module):
    def __init__(self, c, p):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)
        self.norm = nn.GroupNorm(4, c)
        self.fc = nn.Linear(c, c)
        self.p = p

    def forward(self, x):
        y = F.gelu(self.norm(self.conv(x)))
        y = y * torch.sigmoid(self.fc(y.mean(dim=(2, 3))))[:, :, None, None]
        if self.p > 0 and self.training:
            keep = 1.0 - self.p
            y = y * (torch.empty(x.shape[0], device=x.device).bernoulli_(keep) / keep).view(-1, 1, 1, 1)
        return x + y


def gauss(k=11, sigma=1.5):
    d = torch.arange((1 - k) / 2, (1 + k) / 2, 1)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def ssim(p, t, g):
    c1, c2, pad = 0.01 ** 2, 0.03 ** 2, 5
    p = F.pad(p, (pad,) * 4, mode="reflect")
    t = F.pad(t, (pad,) * 4, mode="reflect")
    b, c = p.shape[:2]
    z = torch.cat((p, t, p * p, t * t, p * t))
    gh, gw = g.view(1, 1, 11, 1).repeat(c, 1, 1, 1), g.view(1, 1, 1, 11).repeat(c, 1, 1, 1)
    mu_p, mu_t, e_pp, e_tt, e_pt = F.conv2d(F.conv2d(z, gh, groups=c), gw, groups=c).split(b)
    s_pp, s_tt = (e_pp - mu_p * mu_p).clamp(min=0.0), (e_tt - mu_t * mu_t).clamp(min=0.0)
    s_pt = e_pt - mu_p * mu_t
    smap = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    return smap[..., pad:-pad, pad:-pad].reshape(b, -1).mean(-1).mean()


def hybrid(pred, target, g):
    sq = (pred - target) ** 2
    tmin, tmax = torch.aminmax(target.detach())
    pmin, pmax = torch.aminmax(pred.detach())
    tn = (target - tmin) / (tmax - tmin + 1e-8)
    pn = (pred - pmin) / (pmax - pmin + 1e-8)
    return 0.7 * (0.55 * sq.mean() + 0.45 * (sq * torch.exp(target * 2)).mean()) + 0.3 * (1 - ssim(pn, tn, g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=24)
    ap.add_argument("--droppath", type=int, default=1)
    ap.add_argument("--loss", choices=("hybrid", "mse"), default="hybrid")
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    net = nn.Sequential(nn.Conv2d(5, 16, 3, padding=1), *[Block(16, 0.1 if a.droppath else 0.0) for _ in range(a.blocks)],
                        nn.Conv2d(16, 5, 3, padding=1), nn.Sigmoid()).to(dev).train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.05, fused=True, capturable=True)
    g = gauss().to(dev)
    x, tgt = torch.rand(8, 5, 128, 128, device=dev), torch.rand(8, 5, 128, 128, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(x)
        loss = hybrid(out, tgt, g) if a.loss == "hybrid" else F.mse_loss(out, tgt)
        loss.backward()
        opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    vals = []
    for i in range(6):
        graph.replay()
        vals.append(loss.item())
    print("torch-only PKT=%s blocks=%d droppath=%d loss=%s :" % (os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset"), a.blocks,
                                                               a.droppath, a.loss), " ".join("%.5f" % v for v in vals), flush=True)


if __name__ == "__main__":
    main()
