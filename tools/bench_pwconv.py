"""Per-shape timing of the pointwise-conv kernels (back-to-back launches, HIP events): forward, input gradient, weight gradient
at the shapes of the bench workload.  For A/B runs of kernel variants (KMU_LIB_VARIANT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import _lib
from km_unet_amd.ops import _ptr, _stream

lib = _lib.load()
d = "cuda"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 8
for ci, co, hw in ((16, 64, 128), (64, 16, 128), (16, 48, 128), (16, 16, 128), (32, 128, 64), (128, 32, 64), (32, 96, 64), (64, 256, 32), (256, 64, 32)):
    P = hw * hw
    x = torch.randn(B, ci, P, device=d); w = torch.randn(co, ci, device=d) * 0.1; y = torch.empty(B, co, P, device=d)
    gy = torch.randn(B, co, P, device=d); dx = torch.empty_like(x)
    st = _stream()
    t_f = timeit(lambda: lib.kmu_pwconv_fwd(_ptr(x), _ptr(w), None, _ptr(y), B, ci, co, P, 0, st))
    t_d = timeit(lambda: lib.kmu_pwconv_bwd_input(_ptr(gy), _ptr(w), None, _ptr(dx), B, ci, co, P, 0, st))
    mb = 4.0 * B * P * (ci + co) / 1e6
    print("ci=%3d co=%3d %3dx%-3d  fwd %6.1f us (%4.2f TB/s)   dgrad %6.1f us (%4.2f TB/s)   [%.1f MB]" % (ci, co, hw, hw, t_f, mb / t_f, t_d, mb / t_d, mb), flush=True)
