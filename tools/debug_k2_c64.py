"""Debug probe (GPU): which entries of d_w_bcdt partials are wrong / unwritten for C=64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import km_unet_amd
from km_unet_amd import _lib
from oracle import hsmssd as oh

lib = _lib.load()
for (B, C, Hs) in [(1, 64, 8)]:
    N, L = 64, Hs * Hs
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, L, generator=gen)
    w = [torch.randn(3 * N, C, 1, generator=gen) / C ** 0.5, torch.randn(3 * N, 1, 3, 3, generator=gen) * 0.4,
         torch.randn(2 * C, C, 1, generator=gen) / C ** 0.5, torch.randn(C, C, 1, generator=gen) / C ** 0.5,
         torch.rand(N, generator=gen) * 15 + 1, torch.ones(1) + 0.3]
    gy = torch.randn(B, C, Hs, Hs, generator=gen)
    xo = x.clone().requires_grad_(True)
    wo = [t.clone().requires_grad_(True) for t in w]
    yo, ho = oh.hsmssd(xo, *wo, state_dim=N)
    (yo * gy).sum().backward()
    ref = wo[0].grad.view(3 * N, C)
    d = "cuda"
    xd = x.to(d); wd = [t.to(d).contiguous() for t in w]
    y = torch.empty(B, C, Hs, Hs, device=d); h = torch.empty(B, C, N, device=d)
    state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=d)
    nb = lib.kmu_hsmssd_fwd_ws_bytes(B, C, N, Hs); ws = torch.empty(nb // 4 + 1, device=d)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.kmu_hsmssd_fwd(xd.data_ptr(), wd[0].data_ptr(), wd[1].data_ptr(), wd[2].data_ptr(), wd[3].data_ptr(), wd[5].data_ptr(),
                            y.data_ptr(), h.data_ptr(), state.data_ptr(), ws.data_ptr(), nb, B, C, N, Hs, st)
    assert rc == 0
    P = lib.kmu_hsmssd_bwd_partials(B, C, Hs)
    SENT = 777.0
    pb = torch.full((P, 3 * N, C), SENT, device=d); pdw = torch.full((P, 3 * N, 9), SENT, device=d)
    phz = torch.empty(B, 2 * C, C, device=d); pout = torch.empty(B, C, C, device=d); pD = torch.empty(B, device=d)
    dx = torch.empty_like(xd)
    nb2 = lib.kmu_hsmssd_bwd_ws_bytes(B, C, N, Hs); ws2 = torch.empty(nb2 // 4 + 1, device=d)
    rc = lib.kmu_hsmssd_bwd(xd.data_ptr(), gy.to(d).data_ptr(), None, wd[0].data_ptr(), wd[1].data_ptr(), wd[2].data_ptr(),
                            wd[3].data_ptr(), wd[5].data_ptr(), state.data_ptr(), dx.data_ptr(), pb.data_ptr(), pdw.data_ptr(),
                            phz.data_ptr(), pout.data_ptr(), pD.data_ptr(), ws2.data_ptr(), nb2, B, C, N, Hs, st)
    torch.cuda.synchronize()
    print("case", (B, C, Hs), "rc", rc, "P", P, lib.kmu_last_error())
    pbc = pb.cpu()
    unwritten = (pbc == SENT)
    print("  unwritten entries:", int(unwritten.sum()), "of", pbc.numel())
    got = pbc.sum(0)
    err = (got - ref).abs() / ref.abs().max()
    bad = err > 1e-3
    print("  bad entries:", int(bad.sum()), " bad rows:", sorted(set(bad.nonzero()[:, 0].tolist()))[:40])
    print("  bad cols:", sorted(set(bad.nonzero()[:, 1].tolist()))[:70])
    if bad.any():
        r, c = bad.nonzero()[0].tolist()
        print("  example row %d col %d got %g ref %g ; row slice got %s ref %s" % (r, c, got[r, c], ref[r, c], got[r, :4].tolist(), ref[r, :4].tolist()))
    m = bad.view(3 * N, C // 16, 16).any(-1)
    print("  row:ctile-bad map:", " ".join("%d:%s" % (r, "".join("X" if v else "." for v in m[r].tolist())) for r in range(0, 3 * N)))
    print("  dx err", ((dx.cpu() - xo.grad).abs().max() / xo.grad.abs().max()).item())
