"""Bisect the hipGraph-replay corruption of the captured train step by PROGRAM, not by runtime knob.

    losses : replay the captured step N times and print the loss per replay, with a choice of host-sync pattern
             (--sync item | none | explicit) and of what is inside the graph (--droppath, --loss, --opt)
    trace  : same, but keep every module's forward output and the gradient arriving at it alive inside the graph's
             memory pool and write their L1 norms after every replay to --out (JSON).  Two such files -- one taken with
             DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 and one with =0, same seeds -- are compared with
    diff   : the first (replay, tensor) at which the two runs disagree names the kernel that went wrong.

The environment variable must be set by the caller (the package only sets a default).
"""
import argparse
import json
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")


def build(args):
    import torch
    import km_unet_amd
    from km_unet_amd import train as T
    from km_unet_amd.loss import HybridLoss

    class AtenHybrid(HybridLoss):                      # the tensor-op formulation (what round 1's failing run used)
        part = args.loss

        def _filter(self, x):
            if self.part == "aten_conv":               # MIOpen depthwise convs instead of csrc/gauss11.hip
                c = x.shape[1]
                g = self.gauss.to(x.device)
                gh, gw = g.view(1, 1, self.k, 1).repeat(c, 1, 1, 1), g.view(1, 1, 1, self.k).repeat(c, 1, 1, 1)
                return torch.nn.functional.conv2d(torch.nn.functional.conv2d(x, gh, groups=c), gw, groups=c)
            return super()._filter(x)

        def forward(self, pred, target):
            if self.part == "aten_nominmax":           # SSIM on the raw tensors: no aminmax, no normalisation
                d = pred - target
                return self.alpha * (d * d).mean() + (1 - self.alpha) * (1 - self.ssim(pred, target))
            if self.part == "aten_nossim":             # everything but the SSIM term
                d = pred - target
                sq = d * d
                tmin, tmax = torch.aminmax(target.detach())
                pmin, pmax = torch.aminmax(pred.detach())
                return 0.55 * sq.mean() + 0.45 * (sq * torch.exp(target * 2)).mean() + 1e-3 * ((pred - pmin) / (pmax - pmin + 1e-8)).mean() + 0 * (tmin + tmax)
            d = pred - target
            sq = d * d
            mse = sq.mean()
            weighted = (sq * torch.exp(target * 2)).mean()
            tmin, tmax = torch.aminmax(target.detach())
            pmin, pmax = torch.aminmax(pred.detach())
            tn = (target - tmin) / (tmax - tmin + 1e-8)
            pn = (pred - pmin) / (pmax - pmin + 1e-8)
            ss = self.ssim(pn, tn)
            # the components stay alive in the graph's pool: read back after every replay, they name the node that goes wrong
            self.parts = {"mse": mse.detach(), "wmse": weighted.detach(), "tmin": tmin, "tmax": tmax, "pmin": pmin, "pmax": pmax,
                          "ssim": ss.detach(), "pred_absmax": pred.detach().abs().max(), "pn_absmax": pn.detach().abs().max()}
            return self.alpha * (0.55 * mse + 0.45 * weighted) + (1 - self.alpha) * (1 - ss)

    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=5).cuda().train()
    if not args.droppath:
        for m in model.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
    torch.manual_seed(1234)
    data = torch.rand(args.batch, 10, 1, args.size, args.size, device="cuda")
    st = T.TrainStep(model, data, capturable=True, loss="mse" if args.loss == "mse" else "hybrid")
    if args.loss.startswith("aten"):
        st.criterion = AtenHybrid().cuda()
    if not args.opt:
        st.opt.step = lambda *a, **k: None           # forward + loss + backward only: weights stay fixed
    return torch, model, data, st, T


def run_losses(args):
    torch, model, data, st, T = build(args)
    gs = T.GraphedTrainStep(st, data)
    vals, dev_vals = [], []
    for i in range(args.replays):
        if args.sync == "explicit" and i == 2:
            torch.cuda.current_stream().synchronize()
        loss = gs(data)
        if args.sync == "none":
            dev_vals.append(loss.clone())
        else:
            vals.append(loss.item())
        parts = getattr(st.criterion, "parts", None)
        if parts:
            print("   replay %d parts: " % i + "  ".join("%s=%.5g" % (k, v.item()) for k, v in parts.items()), flush=True)
    if dev_vals:
        vals = [v.item() for v in dev_vals]
    print("PKT=%s droppath=%d loss=%-6s opt=%d sync=%-8s :" % (os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset"), args.droppath,
                                                               args.loss, args.opt, args.sync), " ".join("%.5f" % v for v in vals), flush=True)


def run_trace(args):
    torch, model, data, st, T = build(args)
    fwd, bwd = {}, {}
    for name, mod in model.named_modules():
        def fh(mod, inp, out, name=name):
            o = out[0] if isinstance(out, (tuple, list)) else out
            if not torch.is_tensor(o) or not o.requires_grad:
                return
            fwd[name] = o
            o.register_hook(lambda g, name=name: bwd.__setitem__(name, g))
        mod.register_forward_hook(fh)
    gs = T.GraphedTrainStep(st, data)
    names = ["F:" + n for n in fwd] + ["B:" + n for n in bwd] + ["flat_grad", "flat_param", "loss"]
    tensors = [t.detach() for t in fwd.values()] + [t.detach() for t in bwd.values()] + [st.dp.bucket.flat, st.flat_param.detach()]
    rows = []
    for i in range(args.replays):
        if args.sync == "explicit" and i == 2:
            torch.cuda.current_stream().synchronize()
        loss = gs(data)
        norms = torch._foreach_norm([t.reshape(-1) for t in tensors], 1)
        v = torch.stack([n.double() for n in norms] + [loss.detach().double()]).cpu().tolist()
        rows.append(v)
        print("replay %d loss %.5f" % (i, v[-1]), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({"names": names, "rows": rows, "env": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "unset")}, open(args.out, "w"))


def run_diff(args):
    a, b = json.load(open(args.a)), json.load(open(args.b))
    assert a["names"] == b["names"], "the two traces recorded different tensors"
    names = a["names"]
    for i, (ra, rb) in enumerate(zip(a["rows"], b["rows"])):
        bad = []
        for n, va, vb in zip(names, ra, rb):
            ok = (va == vb) or (abs(va - vb) <= args.tol * max(abs(va), abs(vb), 1e-30))
            if not ok:
                bad.append((n, va, vb))
        print("replay %d: %d of %d tensors differ (tol %.0e)  loss %s=%.5f %s=%.5f" % (i, len(bad), len(names), args.tol, a["env"], ra[-1], b["env"], rb[-1]))
        for n, va, vb in bad[:args.show]:
            print("    %-70s %.6e  vs  %.6e" % (n, va, vb))
        fb = [x for x in bad if x[0].startswith("F:")]
        bb = [x for x in bad if x[0].startswith("B:")]
        if fb:
            print("    first forward tensor that differs :", fb[0][0])
        if bb:
            print("    first backward tensor that differs:", bb[0][0])


def run_selfdiff(args):
    """Replays of a graph whose weights and random numbers are fixed (--opt 0 --droppath 0) must be bit-identical: any tensor
    whose norm changes between replays was written by a kernel with a race (float atomics excepted: DySample / deformable
    conv backward scatter)."""
    a = json.load(open(args.a))
    base = a["rows"][0]
    for i, row in enumerate(a["rows"][1:], start=1):
        bad = [(n, v0, v) for n, v0, v in zip(a["names"], base, row) if v0 != v]
        print("replay %d vs replay 0: %d of %d tensors differ at all" % (i, len(bad), len(a["names"])))
        for n, v0, v in bad[:args.show]:
            print("    %-70s %.9e  vs  %.9e  (rel %.1e)" % (n, v0, v, abs(v - v0) / max(abs(v0), 1e-30)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=("losses", "trace", "diff", "selfdiff"))
    ap.add_argument("a", nargs="?")
    ap.add_argument("b", nargs="?")
    ap.add_argument("--droppath", type=int, default=1)
    ap.add_argument("--loss", choices=("hybrid", "mse", "aten", "aten_conv", "aten_nominmax", "aten_nossim"), default="hybrid")
    ap.add_argument("--opt", type=int, default=1)
    ap.add_argument("--sync", choices=("item", "none", "explicit"), default="explicit")
    ap.add_argument("--replays", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--out", default="gpurun_out/graph_trace.json")
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--show", type=int, default=12)
    args = ap.parse_args()
    {"losses": run_losses, "trace": run_trace, "diff": run_diff, "selfdiff": run_selfdiff}[args.mode](args)


if __name__ == "__main__":
    main()
