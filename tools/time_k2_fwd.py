#!/usr/bin/env python3
"""Device time of the K2 forward (LayerNorm1D + HSMSSD) per launch at the bench shapes, old kernels vs csrc/hsmssd_v2.inc.

    python tools/time_k2_fwd.py [--rows H] [--iters N] [--modes v2,bf16x3]

Each entry point is launched `iters` times back to back on one stream between ONE pair of HIP events (device time per launch
including the dependent-launch boundary, without the per-launch event overhead of bench.py's instrumented steps).  Shapes:
(B, C, Hs) = (8,16,128), (8,32,64), (24,64,32) -- the three levels of KM_UNetV3_SH at B = 8 (the 32x32 level runs its three
direction branches stacked: 24 samples).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd  # noqa: E402
from km_unet_amd import _lib, ops  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--modes", default="v2,bf16x3")
    ap.add_argument("--out", default="")
    ap.add_argument("--stamps", action="store_true", help="print the in-kernel phase stamps of pass 1 (needs an explicit --rows configuration)")
    ap.add_argument("--shape", default="", help="B,C,Hs: only this shape")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    N = 64
    res = {}
    shapes = ((8, 16, 128), (8, 32, 64), (24, 64, 32))
    if args.shape:
        shapes = (tuple(int(v) for v in args.shape.split(",")),)
    for (B, C, Hs) in shapes:
        L = Hs * Hs
        g = torch.Generator().manual_seed(C)
        x = torch.randn(B, C, L, generator=g).to(dev)
        lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        w_bcdt = (torch.randn(3 * N, C, generator=g) / C ** 0.5).to(dev)
        w_dw = (torch.randn(3 * N, 9, generator=g) * 0.4).to(dev)
        w_hz, w_out = (torch.randn(2 * C, C, generator=g) / C ** 0.5).to(dev), (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
        D = torch.ones(1, device=dev)
        y = torch.empty(B, C, Hs, Hs, device=dev)
        h = torch.empty(B, C, N, device=dev)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev)
        xn, stats = torch.empty_like(x), torch.empty(B, L, 2, device=dev)
        wpk = torch.empty(lib.kmu_hsmssd_pack_elems(C, 1), device=dev, dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.kmu_hsmssd_pack_x3(w_bcdt.data_ptr(), w_dw.data_ptr(), wpk.data_ptr(), C, 1, st), "pack")
        row = {}
        if "v2" in args.modes:
            nb = lib.kmu_mixer_fwd_ws_bytes(B, C, N, Hs)
            ws = torch.empty(nb // 4 + 1, device=dev)
            tk = torch.zeros(B, device=dev, dtype=torch.int32)
            lib.kmu_mixer_debug_rows(args.rows)

            def v2(stage, train=True):
                _lib.check(lib.kmu_mixer_fwd_stage(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, w_dw.data_ptr(), w_hz.data_ptr(), w_out.data_ptr(),
                                                   D.data_ptr(), wpk.data_ptr(), y.data_ptr(), h.data_ptr(), state.data_ptr(),
                                                   xn.data_ptr() if train else None, stats.data_ptr() if train else None, ws.data_ptr(), nb,
                                                   tk.data_ptr(), B, C, N, Hs, stage, 1, st), "mixer")
            row["v2_pass1"] = timed(lambda: v2(0), args.iters)
            row["v2_pass2_train"] = timed(lambda: v2(1), args.iters)
            row["v2_pass2_eval"] = timed(lambda: v2(1, False), args.iters)
            row["v2_both_train"] = timed(lambda: (v2(0), v2(1)), args.iters)
            assert int(tk.abs().sum()) == 0
            if args.stamps:
                # phase stamps (100 MHz) of the LAST-ARRIVING workgroup of every sample: [start, staged, main loop done, ticket known, end] and the
                # gate tail's [entry, (m, s) in LDS, factors, acc combined, hpre, hz+gate, out_proj, M pack]
                lib.kmu_mixer_debug_rows((args.rows & 255) | (4 << 8))
                v2(0)
                torch.cuda.synchronize()
                lib.kmu_mixer_debug_rows(args.rows)
                H_, wide_ = args.rows & 15, bool(args.rows & 32)
                rows_ = 4 * H_ * (1 if wide_ or not (args.rows & 16) else 2)
                cols_ = 32 if wide_ else 16
                T_ = -(-Hs // rows_) * -(-Hs // cols_)
                for b in range(min(B, 3)):
                    off = (B * T_ + b) * 128
                    raw = ws[off:off + 128].view(torch.int64).cpu().tolist()
                    t = raw[:5]
                    g_ = raw[16 + 1:16 + 9]
                    print("   sample %d: stage %.2f main %.2f publish %.2f tail %.2f us | tail phases: %s" % (
                        b, (t[1] - t[0]) / 100, (t[2] - t[1]) / 100, (t[3] - t[2]) / 100, (t[4] - t[3]) / 100,
                        " ".join("%.2f" % ((g_[i + 1] - g_[i]) / 100) for i in range(7))))
            lib.kmu_mixer_debug_rows(0)
        if "bf16x3" in args.modes:
            nb = lib.kmu_hsmssd_fwd_ws_bytes(B, C, N, Hs)
            ws = torch.empty(nb // 4 + 1, device=dev)

            def old(stage):
                _lib.check(lib.kmu_hsmssd_fwd_stage_x3_pk(xn.data_ptr(), w_bcdt.data_ptr(), w_dw.data_ptr(), w_hz.data_ptr(), w_out.data_ptr(),
                                                          D.data_ptr(), y.data_ptr(), h.data_ptr(), state.data_ptr(), ws.data_ptr(), nb, B, C, N, Hs,
                                                          stage, 1, wpk.data_ptr(), st), "old")

            def ln():
                _lib.check(lib.kmu_layernorm1d_fwd(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), xn.data_ptr(), stats.data_ptr(), B, C, L, 1e-5, st), "ln")
            row["old_ln"] = timed(ln, args.iters)
            for i, nm in enumerate(("old_pass1", "old_gate", "old_pass2")):
                row[nm] = timed(lambda: old(i), args.iters)
            row["old_all"] = timed(lambda: (ln(), old(0), old(1), old(2)), args.iters)
        res["%d,%d,%d" % (B, C, Hs)] = {k: round(v, 2) for k, v in row.items()}
        print((B, C, Hs), res["%d,%d,%d" % (B, C, Hs)], flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
