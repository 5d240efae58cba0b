#!/usr/bin/env python3
"""VERDICT r2 item 6 (row g1): what would bf16 STORAGE of the streaming glue tensors cost in parity?

The CPU oracle (fp32) is run twice on the same weights and input: as it is, and with every tensor that only streaming kernels touch
-- the input and output of each BatchNorm2d (i.e. the depthwise / pointwise conv outputs feeding it and the normalised tensor), and
their gradients -- rounded to bf16 where the HIP path would store it (fp32 arithmetic in between, as the proposal says).  Printed:
output error, input-gradient and parameter-gradient errors relative to each tensor's maximum (the parity bound is 1e-3).
usage: python tools/bf16_storage_study.py [size]        (CPU only; ~1 min)"""
import copy
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.model import KM_UNetV3 as Oracle, fill_parameters  # noqa: E402


class Round(torch.autograd.Function):
    """value and gradient both stored as bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.manual_seed(0)
    ref = fill_parameters(Oracle(num_classes=5), 2).train()
    for m in ref.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    q = copy.deepcopy(ref)
    for m in q.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.register_forward_pre_hook(lambda mod, inp: (Round.apply(inp[0]),))
            m.register_forward_hook(lambda mod, inp, out: Round.apply(out))
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.rand(2, 5, size, size, generator=g), torch.rand(2, 5, size, size, generator=g)
    res = []
    for net in (ref, q):
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        torch.nn.functional.mse_loss(y, tgt).backward()
        res.append((y.detach(), xi.grad, {n: p.grad for n, p in net.named_parameters() if p.grad is not None}))
    (y0, dx0, g0), (y1, dx1, g1) = res
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    gmax = max(v.abs().max().item() for v in g0.values())
    worst = max(((k, (g1[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-4 * gmax)) for k, v in g0.items()), key=lambda t: t[1])
    n_bad = sum(1 for k, v in g0.items() if (g1[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-4 * gmax) > 1e-3)
    print("bf16 storage of BatchNorm2d inputs / outputs (+ their gradients), fp32 arithmetic, KM_UNetV3_SH train mode [2,5,%d,%d]:" % (size, size))
    print("  output      max |dy|            %.2e   (bound 1e-3)" % (y1 - y0).abs().max().item())
    print("  d input     rel. to max         %.2e" % rel(dx1, dx0))
    print("  parameters  worst rel. to max   %.2e  (%s); %d of %d tensors beyond 1e-3" % (worst[1], worst[0], n_bad, len(g0)))


if __name__ == "__main__":
    main()
