"""Development aid: run the K2 v2 forward (csrc/hsmssd_v2.inc) under a forced pass-1 configuration and compare state / outputs with the default."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd
from km_unet_amd import _lib
lib = _lib.load()


def fwd(B, C, Hs, cfg, x, lw, lb, w_bcdt, w_dw, w_hz, w_out, D, wpk):
    N, L, dev = 64, Hs * Hs, x.device
    y, h = torch.empty(B, C, Hs, Hs, device=dev), torch.empty(B, C, N, device=dev)
    state = torch.zeros(lib.kmu_hsmssd_state_elems(B, C, N), device=dev)
    nb = lib.kmu_mixer_fwd_ws_bytes(B, C, N, Hs)
    ws = torch.zeros(nb // 4 + 1, device=dev)
    tk = torch.zeros(B, device=dev, dtype=torch.int32)
    st = torch.cuda.current_stream().cuda_stream
    lib.kmu_mixer_debug_rows(cfg)
    for stage in (0, 1):
        _lib.check(lib.kmu_mixer_fwd_stage(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, w_dw.data_ptr(), w_hz.data_ptr(), w_out.data_ptr(), D.data_ptr(),
                                           wpk.data_ptr(), y.data_ptr(), h.data_ptr(), state.data_ptr(), None, None, ws.data_ptr(), nb, tk.data_ptr(), B, C, N, Hs,
                                           stage, 1, st), "mixer")
    lib.kmu_mixer_debug_rows(0)
    torch.cuda.synchronize()
    return y, h, state.view(B, -1), ws


def main():
    B, C, Hs = [int(a) for a in sys.argv[1:4]]
    cfgs = [int(a) for a in sys.argv[4:]]
    g = torch.Generator().manual_seed(1)
    N = 64
    x = torch.randn(B, C, Hs * Hs, generator=g).cuda()
    lw, lb = torch.ones(C).cuda(), torch.zeros(C).cuda()
    w_bcdt, w_dw = (torch.randn(3 * N, C, generator=g) / C ** 0.5).cuda(), (torch.randn(3 * N, 9, generator=g) * 0.4).cuda()
    w_hz, w_out, D = (torch.randn(2 * C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.ones(1).cuda()
    wpk = torch.empty(lib.kmu_hsmssd_pack_elems(C, 1), device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.kmu_hsmssd_pack_x3(w_bcdt.data_ptr(), w_dw.data_ptr(), wpk.data_ptr(), C, 1, torch.cuda.current_stream().cuda_stream), "pack")
    ref = fwd(B, C, Hs, 0, x, lw, lb, w_bcdt, w_dw, w_hz, w_out, D, wpk)
    for cfg in cfgs:
        for rep in range(3):
            y, h, st, ws = fwd(B, C, Hs, cfg, x, lw, lb, w_bcdt, w_dw, w_hz, w_out, D, wpk)
            rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
            M, S, hp = st[:, :64], st[:, 64:128], st[:, 128:128 + C * 64]
            rM, rS, rhp = ref[2][:, :64], ref[2][:, 64:128], ref[2][:, 128:128 + C * 64]
            print("cfg %d rep %d: y %.1e h %.1e | M %.1e S %.1e hpre %.1e | bad M cols %s" % (cfg, rep, rel(y, ref[0]), rel(h, ref[1]), rel(M, rM), rel(S, rS), rel(hp, rhp),
                  ((M - rM).abs() > 1e-4).nonzero()[:6].tolist()))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "partials"):
    main()


def partials(ws, B, C, Hs, rows):
    tx = (Hs + 15) // 16
    T = tx * ((Hs + rows - 1) // rows)
    T1 = tx * ((Hs + 3) // 4)
    ms = ws[:B * T * 128].view(B, T, 2, 64)
    acc = ws[B * T1 * 128:B * T1 * 128 + B * T * C * 64].view(B, T, C, 64)
    return ms, acc, tx


def compare_partials():
    B, C, Hs = [int(a) for a in sys.argv[2:5]]
    g = torch.Generator().manual_seed(1)
    N = 64
    x = torch.randn(B, C, Hs * Hs, generator=g).cuda()
    lw, lb = torch.ones(C).cuda(), torch.zeros(C).cuda()
    w_bcdt, w_dw = (torch.randn(3 * N, C, generator=g) / C ** 0.5).cuda(), (torch.randn(3 * N, 9, generator=g) * 0.4).cuda()
    w_hz, w_out, D = (torch.randn(2 * C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), torch.ones(1).cuda()
    wpk = torch.empty(lib.kmu_hsmssd_pack_elems(C, 1), device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.kmu_hsmssd_pack_x3(w_bcdt.data_ptr(), w_dw.data_ptr(), wpk.data_ptr(), C, 1, torch.cuda.current_stream().cuda_stream), "pack")
    a = fwd(B, C, Hs, 18, x, lw, lb, w_bcdt, w_dw, w_hz, w_out, D, wpk)      # 16-row tiles (8 waves)
    b = fwd(B, C, Hs, 2, x, lw, lb, w_bcdt, w_dw, w_hz, w_out, D, wpk)       # 8-row tiles
    msA, accA, tx = partials(a[3], B, C, Hs, 16)
    msB, accB, _ = partials(b[3], B, C, Hs, 8)
    for ty in range(msA.shape[1] // tx):
        for t in range(tx):
            tA, t0, t1 = ty * tx + t, (2 * ty) * tx + t, (2 * ty + 1) * tx + t
            m0, s0, m1, s1 = msB[:, t0, 0], msB[:, t0, 1], msB[:, t1, 0], msB[:, t1, 1]
            m = torch.maximum(m0, m1)
            f0, f1 = torch.exp(m0 - m), torch.exp(m1 - m)
            s = s0 * f0 + s1 * f1
            acc = accB[:, t0] * f0[:, None, :] + accB[:, t1] * f1[:, None, :]
            dm, ds = (msA[:, tA, 0] - m).abs().max().item(), ((msA[:, tA, 1] - s).abs().max() / s.abs().max()).item()
            da = (accA[:, tA] - acc).abs() / acc.abs().max()
            bad = (da > 1e-3).nonzero()
            print("tile", tA, "dm %.1e ds %.1e dacc %.1e" % (dm, ds, da.max().item()), "bad entries", bad.shape[0], "c values", sorted(set(bad[:, 1].tolist()))[:40],
                  "n values", sorted(set(bad[:, 2].tolist()))[:40])


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "partials":
    compare_partials()
