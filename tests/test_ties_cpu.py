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
