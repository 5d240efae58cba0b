"""Where does the whole-model input gradient of the HIP model leave the oracle's?  

For one smoke()-style input (seed on the command line) run three models with identical weights -- the HIP model on
cuda:0, the fp32 CPU oracle, and an fp64 copy of the oracle as arbiter -- and record, for every module the models
share by name, the forward output and the gradient arriving at that output.  Printed in forward order:

    name  shape  |  fwd: gpu-vs-f64, o32-vs-f64  |  grad-at-output: gpu-vs-f64, o32-vs-f64

(max |diff| / max |ref|).  Walking the table from the bottom (= backward order) the first row whose gradient error
jumps names the module whose backward produced it; its inputs / outputs / gradients are saved to --dump for offline
analysis on the CPU (kink search).  The oracle is used as the checker only; nothing here is product code.

    python tools/grad_divergence.py --seed 3 [--train] [--dump gpurun_out/graddiv]
"""
import argparse
import copy
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch  # noqa: E402


def trace(model, x, tgt, loss_fn):
    recs, order, hooks = {}, [], []

    def first_tensor(o):
        if torch.is_tensor(o):
            return o
        if isinstance(o, (tuple, list)):
            for t in o:
                if torch.is_tensor(t):
                    return t
        return None

    for name, mod in model.named_modules():
        def fh(mod, inp, out, name=name):
            o = first_tensor(out)
            if o is None or not o.requires_grad or name in recs:
                return
            r = recs[name] = {"out": o.detach().double().cpu().clone()}
            i = first_tensor(inp)
            if i is not None:
                r["in"] = i.detach().double().cpu().clone()
            order.append(name)
            o.register_hook(lambda g, r=r: r.__setitem__("gout", g.detach().double().cpu().clone()))
        hooks.append(mod.register_forward_hook(fh))
    xr = x.clone().requires_grad_(True)
    y = model(xr)
    loss = loss_fn(y, tgt)
    loss.backward()
    for h in hooks:
        h.remove()
    recs["<input>"] = {"out": xr.detach().double().cpu(), "gout": xr.grad.detach().double().cpu()}
    return recs, ["<input>"] + order, loss.item()


def rel(a, b):
    if a is None or b is None or a.shape != b.shape:
        return float("nan")
    den = b.abs().max().item()
    return ((a - b).abs().max().item() / den) if den > 0 else (a - b).abs().max().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--train", action="store_true", help="train mode (BatchNorm batch statistics; DropPath disabled)")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dump", default=None)
    ap.add_argument("--thresh", type=float, default=1e-4)
    args = ap.parse_args()

    import km_unet_amd
    from oracle.model import KM_UNetV3 as Oracle, fill_parameters
    o32 = fill_parameters(Oracle(num_classes=5), 2)
    for m in o32.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    o32 = o32.train() if args.train else o32.eval()
    o64 = copy.deepcopy(o32).double()
    gm = km_unet_amd.KM_UNetV3(num_classes=5)
    gm.load_state_dict(o32.state_dict(), strict=True)
    for m in gm.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    gm = gm.to("cuda:0")
    gm = gm.train() if args.train else gm.eval()

    g = torch.Generator().manual_seed(args.seed)
    x = torch.rand(args.batch, 5, args.size, args.size, generator=g)
    tgt = torch.rand(args.batch, 5, args.size, args.size, generator=g)
    mse = torch.nn.functional.mse_loss
    r64, order, l64 = trace(o64, x.double(), tgt.double(), mse)
    r32, _, l32 = trace(o32, x, tgt, mse)
    rg, order_g, lg = trace(gm, x.to("cuda:0"), tgt.to("cuda:0"), mse)
    torch.cuda.synchronize()
    print("loss  f64 %.9f  o32 %.9f  gpu %.9f" % (l64, l32, lg))
    print("%-58s %-18s | %9s %9s | %9s %9s" % ("module (forward order)", "shape", "fwd gpu", "fwd o32", "gout gpu", "gout o32"))
    rows = []
    for name in order:
        if name not in rg or name not in r64:
            continue
        a, b, c = rg[name], r32.get(name, {}), r64[name]
        row = (name, tuple(c["out"].shape), rel(a.get("out"), c["out"]), rel(b.get("out"), c["out"]),
               rel(a.get("gout"), c.get("gout")), rel(b.get("gout"), c.get("gout")))
        rows.append(row)
        print("%-58s %-18s | %9.2e %9.2e | %9.2e %9.2e" % row)
    # first (in backward order) module whose output gradient is fine while the gradient at the NEXT recorded
    # tensor upstream (earlier in forward order) is not
    bad = [r for r in rows if r[4] == r[4] and r[4] > args.thresh]
    print("\nrows with gout gpu-vs-f64 > %.0e: %d of %d" % (args.thresh, len(bad), len(rows)))
    if bad:
        last_bad = bad[-1][0]
        print("deepest (latest in forward order) tensor with a wrong gradient:", last_bad)
    if args.dump:
        os.makedirs(args.dump, exist_ok=True)
        # everything that is needed to redo the analysis offline: the input, and for the 12 deepest bad rows the
        # module's input / output / output-gradient from the GPU and the fp64 oracle
        pack = {"x": x, "tgt": tgt, "rows": rows}
        for name in [r[0] for r in bad[-12:]]:
            pack[name] = {"gpu": {k: v.float() for k, v in rg[name].items()}, "f64": r64[name]}
        torch.save(pack, os.path.join(args.dump, "graddiv_seed%d%s.pt" % (args.seed, "_train" if args.train else "")))


if __name__ == "__main__":
    main()
