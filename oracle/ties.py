"""Discontinuity ties between two correct fp32 implementations (TEST INFRASTRUCTURE, oracle side only).

KM-UNet is piecewise smooth: ~2e6 ReLU pre-activations per forward at [2,5,32,32] (StableHybridKANConv.post_act
KM_UNetV3_SH.py:94, the FFN's ConvLayer2D act vim_utils_init.py:62-89, LocalContrastAttention.fc, DAGEM's MLPs).  An
fp32 sum of O(1) terms carries ~1e-7 of rounding, so with that many elements roughly one input in seven has a
pre-activation closer to 0 than the rounding of its own summation.  Two correct implementations that round differently
then take different branches THERE: the forward moves by 1e-7, but the gradient of that element switches on or off,
and d loss / d input changes by ~1e-2 of its maximum in that element's cone.  (B-spline knots are not such points: the
cubic basis is C2; DySample's floor() only affects offset gradients and sits 0.25 px from the nearest integer at init.)

This module makes that statement checkable instead of asserted:
  relu_near_ties(model, x)          -> the ReLU elements of `model` (an oracle) whose |pre-activation| is within `rel`
                                       of the layer's largest, i.e. whose branch fp32 rounding can decide
  grad_with_flips(model, x, loss, flips) -> d loss / d x with the listed elements' branches inverted
  explain_by_ties(...)              -> is  dx_other - dx_oracle  a {0,1}-combination of those single-element flips?
"""
import torch
import torch.nn as nn


class _FlipReLU(torch.autograd.Function):
    """relu(x) whose backward mask is (x > 0) XOR flip."""

    @staticmethod
    def forward(ctx, x, flip):
        ctx.save_for_backward((x > 0) ^ flip)
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return g * m, None


def _relus(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, nn.ReLU)]


def relu_near_ties(model, x, rel=3e-6):
    """[(module name, flat index, pre-activation, layer max)] for every ReLU input element with |pre| <= rel * max|pre|."""
    found, hooks = [], []
    for name, mod in _relus(model):
        def pre(mod, inp, name=name):
            p = inp[0].detach()
            lim = rel * p.abs().max().item()
            idx = (p.abs().flatten() <= lim).nonzero().flatten()
            for i in idx.tolist():
                found.append((name, i, p.flatten()[i].item(), lim / rel))
        hooks.append(mod.register_forward_pre_hook(pre))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return found


def grad_with_flips(model, x, loss_fn, flips=(), params=False):
    """Gradients of loss_fn(model(x)) with the ReLU branches of `flips` = [(module name, flat index), ...] inverted.
    -> d/dx, or with params=True the dict {"<input>": d/dx, parameter name: d/dp for every parameter that receives one}."""
    by_mod = {}
    for name, idx in flips:
        by_mod.setdefault(name, []).append(idx)
    saved = []
    for name, mod in _relus(model):
        if name in by_mod:
            def fwd(inp, idxs=by_mod[name]):
                flip = torch.zeros(inp.numel(), dtype=torch.bool)
                flip[idxs] = True
                return _FlipReLU.apply(inp, flip.view_as(inp))
            saved.append((mod, mod.forward))
            mod.forward = fwd
    try:
        for q in model.parameters():
            q.grad = None
        xr = x.clone().requires_grad_(True)
        loss_fn(model(xr)).backward()
    finally:
        for mod, f in saved:
            mod.forward = f
    if not params:
        return xr.grad.detach()
    out = {"<input>": xr.grad.detach()}
    out.update({n: q.grad.detach().clone() for n, q in model.named_parameters() if q.grad is not None})
    return out


def explain_by_ties(model, x, loss_fn, other, rel=3e-6, tol=1e-4, max_ties=96):
    """Are another implementation's gradients the oracle's up to branch flips at near-tie ReLU elements?

    `other`: its d loss / d x (a tensor), or a dict {"<input>": d/dx, parameter name: gradient, ...} as grad_with_flips(params=True)
    returns.  Every tensor is measured relative to the max magnitude of the oracle's own (floored at 1e-4 of the largest
    one: a parameter whose gradient is pure rounding noise cannot be compared in relative terms).  HSMSSD.A is skipped: its
    true gradient is exactly zero (softmax shift invariance), implementations return 0 or ~1e-9 of noise.
    -> (ok, report); ok: the residual  other - oracle  is, to `tol`, a combination  sum_i c_i Delta_i  of the single-flip
    gradient changes Delta_i of the near-tie elements with c_i in {0, 1} (least-squares fit, rounded; flips are first-order
    additive: two of them interact only through the product of two already tiny pathways)."""
    many = isinstance(other, dict)
    ref = grad_with_flips(model, x, loss_fn, params=many)
    if many:
        keys = [k for k in ref if not k.endswith(".A")]
        missing = [k for k in keys if k not in other]
        assert not missing, "gradients missing from the other implementation: %s" % missing[:5]
        other = [other[k].detach().cpu().double() for k in keys]
        ref = [ref[k] for k in keys]
    else:
        keys, other, ref = ["<input>"], [other.detach().cpu().double()], [ref]
    assert all(a.shape == b.shape for a, b in zip(ref, other)), "gradient shapes do not match"
    gmax = max(r.abs().max().item() for r in ref)
    inv = [1.0 / max(r.abs().max().item(), 1e-4 * gmax, 1e-30) for r in ref]

    def flat(ts):
        if isinstance(ts, dict):
            ts = [ts[k] for k in keys]
        elif torch.is_tensor(ts):
            ts = [ts]
        return torch.cat([t.double().flatten() * s for t, s in zip(ts, inv)])

    f0 = flat(ref)
    res = flat(other) - f0
    report = {"err_before": res.abs().max().item(), "ties": []}
    report["err_after"] = report["err_before"]
    if report["err_before"] <= tol:
        return True, report
    ties = relu_near_ties(model, x, rel)
    report["n_ties"] = len(ties)
    if not ties or len(ties) > max_ties:
        return False, report
    deltas = []
    for n, i, _, _ in ties:
        deltas.append(flat(grad_with_flips(model, x, loss_fn, [(n, i)], params=many)) - f0)
    # a flip whose own effect is below tol/5 cannot be told from rounding noise and cannot explain anything: leave it out
    # of the fit (it would only make the least-squares problem ill-conditioned)
    keep = [k for k, d in enumerate(deltas) if d.abs().max().item() > 0.2 * tol]
    report["n_fitted"] = len(keep)
    after = res
    if keep:
        A = torch.stack([deltas[k] for k in keep], dim=1)
        sol = torch.linalg.lstsq(A, after.unsqueeze(1)).solution.flatten()
        coef = sol.round().clamp(0, 1)                             # a flip either happened or it did not
        after = after - A @ coef
        report["ties"] = [{"module": ties[k][0], "index": ties[k][1], "pre": ties[k][2], "layer_max": ties[k][3],
                           "fit": c.item(), "flipped": bool(r.item())} for k, c, r in zip(keep, sol, coef)]
    report["err_after"] = after.abs().max().item()
    return report["err_after"] <= tol, report


class _MaskReLU(torch.autograd.Function):
    """relu(x) whose backward uses a given branch mask."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return g * m, None


def collect_gpu_relu_masks(gpu_model, run):
    """Run `run()` (a forward of the HIP model) and return the ReLU branch masks it took, in execution order: the fused ops
    report theirs through km_unet_amd.ops.RELU_TAP, plain nn.ReLU modules (DAGEM) through forward hooks."""
    from km_unet_amd import ops
    masks, hooks = [], []
    for mod in gpu_model.modules():
        if isinstance(mod, nn.ReLU):
            hooks.append(mod.register_forward_hook(lambda m, i, o: masks.append((o.detach() > 0).cpu())))
    ops.RELU_TAP = masks
    try:
        out = run()
    finally:
        ops.RELU_TAP = None
        for h in hooks:
            h.remove()
    return out, masks


def grad_with_masks(model, x, loss_fn, masks, params=False):
    """Gradients of loss_fn(model(x)) with the k-th ReLU call of the forward differentiating through masks[k] (another
    implementation's branch decisions).  -> (gradients as grad_with_flips, flips) where flips lists, per ReLU call with a
    disagreement, (module name, number of flipped elements, max |pre-activation| / layer max among them)."""
    calls, flips, saved = [0], [], []
    for name, mod in _relus(model):
        def fwd(inp, name=name):
            k = calls[0]
            calls[0] += 1
            if k >= len(masks) or masks[k].numel() != inp.numel():
                raise RuntimeError("ReLU call %d (%s, %s) has no matching mask (%s)" % (
                    k, name, tuple(inp.shape), "none left" if k >= len(masks) else tuple(masks[k].shape)))
            m = masks[k].reshape(inp.shape)
            diff = m != (inp.detach() > 0)
            if bool(diff.any()):
                p = inp.detach().abs()
                flips.append((name, int(diff.sum()), (p[diff].max() / p.max()).item()))
            return _MaskReLU.apply(inp, m)
        saved.append((mod, mod.forward))
        mod.forward = fwd
    try:
        for q in model.parameters():
            q.grad = None
        xr = x.clone().requires_grad_(True)
        loss_fn(model(xr)).backward()
    finally:
        for mod, f in saved:
            mod.forward = f
    if calls[0] != len(masks):
        raise RuntimeError("the oracle made %d ReLU calls, the other implementation reported %d masks" % (calls[0], len(masks)))
    if not params:
        return xr.grad.detach(), flips
    out = {"<input>": xr.grad.detach()}
    out.update({n: q.grad.detach().clone() for n, q in model.named_parameters() if q.grad is not None})
    return out, flips


def explain_by_masks(model, x, loss_fn, other, masks, tie_rel=2e-4, tol=2e-4):
    """The direct form of the tie allowance: differentiate the (fp64) oracle through the OTHER implementation's own ReLU
    branches (`masks`, from collect_gpu_relu_masks) and compare.  ok iff (a) every branch that differs from the oracle's own
    sits at a pre-activation within `tie_rel` of zero relative to its layer's largest (i.e. within what the other
    implementation's rounding can move: ~1e-7 for exact-fp32 kernels, ~1e-5 for the split-bf16 matrix-core kernels), and
    (b) with those branches the gradients agree to `tol` (per tensor, relative to its own maximum floored at 1e-4 of the
    largest tensor's: a bias in front of a batch-statistics BatchNorm has an exactly-zero gradient and holds only noise).
    -> (ok, report)"""
    many = isinstance(other, dict)
    ref, flips = grad_with_masks(model, x, loss_fn, masks, params=many)
    if many:
        keys = [k for k in ref if not k.endswith(".A")]
        missing = [k for k in keys if k not in other]
        assert not missing, "gradients missing from the other implementation: %s" % missing[:5]
        pairs = [(k, ref[k], other[k].detach().cpu().double()) for k in keys]
    else:
        pairs = [("<input>", ref, other.detach().cpu().double())]
    gmax = max(r.abs().max().item() for _, r, _ in pairs)
    worst = ("", 0.0)
    for k, r, o in pairs:
        e = (o - r.double()).abs().max().item() / max(r.abs().max().item(), 1e-4 * gmax, 1e-30)
        if e > worst[1]:
            worst = (k, e)
    far = [f for f in flips if f[2] > tie_rel]
    report = {"flips": flips, "n_flipped": sum(f[1] for f in flips), "max_flip_rel": max([f[2] for f in flips] + [0.0]),
              "err": worst[1], "worst": worst[0], "not_ties": far}
    return (not far) and worst[1] <= tol, report


def describe_masks(report):
    return "%d ReLU branch flip(s) in %d layer(s), furthest pre-activation %.1e of its layer max; gradients with the same branches: %.2e (%s)%s" % (
        report["n_flipped"], len(report["flips"]), report["max_flip_rel"], report["err"], report["worst"],
        "; NOT TIES: %s" % report["not_ties"][:3] if report["not_ties"] else "")


def describe(report):
    fl = [t for t in report.get("ties", []) if t["flipped"]]
    return "err %.2e -> %.2e after %d flip(s) among %d near-tie ReLU elements%s" % (
        report["err_before"], report["err_after"], len(fl), report.get("n_ties", 0),
        "".join("; %s[%d] pre=%.2e (layer max %.1f)" % (t["module"], t["index"], t["pre"], t["layer_max"]) for t in fl))
