"""oracle/ties.py (the ReLU-tie allowance used by smoke() and the whole-model gradient tests) on the CPU: planted
single-element branch flips are recovered exactly, a genuine gradient error is NOT explained away."""
import copy

import torch

from oracle import ties
from oracle.model import KM_UNetV3 as Oracle, fill_parameters


def test_planted_flips_are_found_and_real_errors_are_not():
    o32 = fill_parameters(Oracle(num_classes=5), 2).eval()
    o64 = copy.deepcopy(o32).double()
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.rand(1, 5, 16, 16, generator=g), torch.rand(1, 5, 16, 16, generator=g)
    loss32 = lambda y: torch.nn.functional.mse_loss(y, tgt)
    loss64 = lambda y: torch.nn.functional.mse_loss(y, tgt.double())
    near = ties.relu_near_ties(o64, x.double(), rel=2e-5)
    assert len(near) >= 2
    dx0 = ties.grad_with_flips(o64, x.double(), loss64)
    # plant the two near-tie flips with the largest effect on dx into the fp32 oracle = "another correct implementation"
    eff = sorted(((ties.grad_with_flips(o64, x.double(), loss64, [(n, i)]) - dx0).abs().max().item(), n, i) for n, i, _, _ in near)
    planted = [(n, i) for _, n, i in eff[-2:]]
    assert eff[-2][0] > 1e-3 * dx0.abs().max().item(), "fixture too tame: no near-tie flip matters at this input"
    other = ties.grad_with_flips(o32, x, loss32, planted)
    ok, rep = ties.explain_by_ties(o64, x.double(), loss64, other, rel=2e-5, tol=1e-4)
    assert ok and rep["err_before"] > 1e-3, rep
    assert sorted((t["module"], t["index"]) for t in rep["ties"] if t["flipped"]) == sorted(planted)
    # also with parameter gradients in the comparison
    other_all = ties.grad_with_flips(o32, x, loss32, planted, params=True)
    ok, rep = ties.explain_by_ties(o64, x.double(), loss64, other_all, rel=2e-5, tol=1e-3)
    assert ok, ties.describe(rep)
    # a 5 % error in a patch of the gradient is not a tie
    bad = other.clone()
    bad[0, 2, 4:8, 4:8] *= 1.05
    ok, rep = ties.explain_by_ties(o64, x.double(), loss64, bad, rel=2e-5, tol=1e-4)
    assert not ok


def _oracle_masks(model, x):
    import torch.nn as nn
    masks, hooks = [], []
    for mod in model.modules():
        if isinstance(mod, nn.ReLU):
            hooks.append(mod.register_forward_pre_hook(lambda m, i: masks.append(i[0].detach() > 0)))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return masks


def test_mask_transplant_accepts_tie_flips_and_rejects_the_rest():
    """explain_by_masks: the fp64 oracle differentiated through another implementation's ReLU branches.  Branches flipped at
    near-tie elements are accepted and reproduce that implementation's gradient; a flip far from zero, or a gradient error
    that is not a branch effect, is refused."""
    o32 = fill_parameters(Oracle(num_classes=5), 2).eval()
    o64 = copy.deepcopy(o32).double()
    g = torch.Generator().manual_seed(1)
    x, tgt = torch.rand(1, 5, 16, 16, generator=g), torch.rand(1, 5, 16, 16, generator=g)
    loss32 = lambda y: torch.nn.functional.mse_loss(y, tgt)
    loss64 = lambda y: torch.nn.functional.mse_loss(y, tgt.double())
    near = ties.relu_near_ties(o64, x.double(), rel=2e-5)
    names = [n for n, m in o64.named_modules() if isinstance(m, torch.nn.ReLU)]
    order = []                                             # ReLU call order (module names) of one forward
    hooks = [m.register_forward_pre_hook(lambda mod, i, n=n: order.append(n)) for n, m in o64.named_modules() if n in names]
    with torch.no_grad():
        o64(x.double())
    for h in hooks:
        h.remove()
    planted = [(near[0][0], near[0][1]), (near[-1][0], near[-1][1])]
    other = ties.grad_with_flips(o32, x, loss32, planted)           # "another implementation" that took the other branch twice
    masks = _oracle_masks(o32, x)
    for n, i in planted:
        m = masks[order.index(n)].view(-1)
        m[i] = ~m[i]
    ok, rep = ties.explain_by_masks(o64, x.double(), loss64, other, masks, tie_rel=1e-4, tol=1e-4)
    assert ok and rep["n_flipped"] >= 2, ties.describe_masks(rep)
    # the same gradient but reported WITHOUT the flips: the oracle's own branches do not reproduce it
    ok, rep = ties.explain_by_masks(o64, x.double(), loss64, other, _oracle_masks(o32, x), tie_rel=1e-4, tol=1e-4)
    assert not ok or rep["err"] <= 1e-4        # (if both planted flips happen to be negligible the gradient simply agrees)
    # a flip far from zero is not a tie, whatever the gradients do
    masks = _oracle_masks(o32, x)
    big = masks[0].view(-1)
    pre0 = []
    h = dict(o64.named_modules())[order[0]].register_forward_pre_hook(lambda mod, i: pre0.append(i[0].detach().abs().flatten()))
    with torch.no_grad():
        o64(x.double())
    h.remove()
    j = int(pre0[0].argmax())
    big[j] = ~big[j]
    ok, rep = ties.explain_by_masks(o64, x.double(), loss64, other, masks, tie_rel=1e-4, tol=1.0)
    assert not ok and rep["not_ties"]
