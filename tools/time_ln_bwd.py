import sys, torch
sys.path.insert(0, '/root/repo')
import km_unet_amd
from km_unet_amd import _lib
lib = _lib.load()
for (B, C, L) in ((8, 16, 16384), (8, 32, 4096), (24, 64, 1024)):
    x = torch.randn(B, C, L, device='cuda'); dy = torch.randn_like(x); ad = torch.randn_like(x); dx = torch.empty_like(x)
    w = torch.ones(C, device='cuda'); stats = torch.rand(B, L, 2, device='cuda') + 0.5
    rows = lib.kmu_layernorm1d_partials(B, C, L)
    dwp = torch.empty(rows, C, device='cuda'); dbp = torch.empty(rows, C, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.kmu_layernorm1d_bwd_add(x.data_ptr(), w.data_ptr(), stats.data_ptr(), dy.data_ptr(), ad.data_ptr(), dx.data_ptr(), dwp.data_ptr(), dbp.data_ptr(), B, C, L, 1, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): f()
    e.record(); torch.cuda.synchronize()
    print((B, C, L), 'ln bwd %.1f us' % (s.elapsed_time(e) / 50 * 1e3), 'rows', rows, '%.2f TB/s' % (4 * B * C * L * 4 / (s.elapsed_time(e) / 50 * 1e-3) / 1e12))
