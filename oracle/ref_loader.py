"""Container-only importer for the upstream reference (TEST INFRASTRUCTURE).

The reference lives read-only at /root/reference and does NOT exist on the GPU
box, so nothing under tests/ marked ``gpu``, ``smoke()`` or ``bench.py`` may
call this.  It is used only by ``tests/golden/make_golden.py`` (fixture
generation) and by the CPU tests that re-validate the oracle while the
reference is still mounted.

Third-party packages the reference imports but never executes on the hot path
(timm, fvcore, pywt, torchvision) are absent from the image; they are replaced
by minimal stand-ins *for import purposes only* (SURVEY.md Appendix B):

* ``timm.layers.trunc_normal_`` / ``timm.models.register_model`` /
  ``fvcore.nn.flop_count`` / ``timm.layers.SqueezeExcite`` -- never called by
  KM-UNet's forward (efficient_vim_init.py:7-9, vim_utils_init.py:3).
* ``timm.models.layers.DropPath`` -- per-sample Bernoulli(1-p)/(1-p) mask in
  train mode, identity in eval (timm 0.9.16 semantics); parity fixtures run in
  eval mode so it never fires.
* ``pywt.Wavelet('haar')`` -- only ``rec_lo``/``rec_hi`` taps are read
  (WPL/iwp.py:50-52); these are the mathematical constants +-1/sqrt(2).
* ``torchvision.ops.DeformConv2d`` -- bound to oracle.deform.DeformConv2d, our
  restatement of torchvision 0.14 semantics => anything that flows through it
  is "parity unpinned" (DESIGN.md).
"""
import math
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("KMUNET_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "KM_UNetV3_SH.py"))


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class _Empty(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def _install_stubs():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_kmunet_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ident = lambda fn=None, *a, **k: fn
    layers = mod("timm.layers", trunc_normal_=nn.init.trunc_normal_, SqueezeExcite=_Empty,
                 DropPath=_DropPath)
    mlayers = mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=_DropPath)
    models = mod("timm.models", register_model=ident, layers=mlayers)
    mod("timm", layers=layers, models=models, _kmunet_stub=True)
    fnn = mod("fvcore.nn", flop_count=None)
    mod("fvcore", nn=fnn)

    s = 1.0 / math.sqrt(2.0)

    class Wavelet:
        def __init__(self, name):
            assert name == "haar", "only the haar taps are restated"
            self.rec_lo = [s, s]
            self.rec_hi = [s, -s]

    mod("pywt", Wavelet=Wavelet)

    from oracle.deform import DeformConv2d  # our restatement (parity unpinned)
    ops = mod("torchvision.ops", DeformConv2d=DeformConv2d)
    mod("torchvision", ops=ops)


def load():
    """Import the reference modules; returns a namespace of the classes used."""
    if not available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # reference dir is read-only
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    ns.kanconv = importlib.import_module("convKAN.KANConv2Dlayers")
    ns.kanlayers = importlib.import_module("convKAN.KANlayers")
    ns.vim = importlib.import_module("vim_block_init.efficient_vim_init")
    ns.vim_utils = importlib.import_module("vim_block_init.vim_utils_init")
    ns.dysample = importlib.import_module("DySample_md")
    ns.iwp = importlib.import_module("WPL.iwp")
    ns.dagem = importlib.import_module("DAGEM_md")
    ns.sh = importlib.import_module("KM_UNetV3_SH")
    ns.laps = importlib.import_module("KM_UNetV3_LAPS")
    return ns
