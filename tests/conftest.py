import os
import sys
import warnings

os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")   # same numerics policy as km-unet_amd/__init__.py
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # same hipGraph policy as km-unet_amd/__init__.py

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore", category=UserWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is a hard skip, never a silent CPU pass.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiu" else a
    return out


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny): the north_star's "1e-3 rel fp32" measure."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def outlier_fraction(a, b, tol=1e-3):
    """fraction of elements further than tol * max|b| from the reference"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs() > tol * b.abs().max().clamp_min(1e-30)).double().mean().item()


@pytest.fixture
def golden():
    return load_golden
