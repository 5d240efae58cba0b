#!/usr/bin/env python3
"""Device time of the K2 backward per launch at the bench shapes: the round-4 path (csrc/hsmssd_bwdc.inc: correlation, C-row
contractions, gate, pass B on the {B, dt} rows) against the older one (pass A on the matrix core, gate, pass B on all 192 rows).

    python tools/time_k2_bwd.py [--iters N] [--shape B,C,Hs] [--out file.json]

Each entry point is launched `iters` times back to back on one stream between ONE pair of HIP events, as tools/time_k2_fwd.py.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import km_unet_amd  # noqa: E402,F401
from km_unet_amd import _lib  # noqa: E402
from time_k2_fwd import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--shape", default="", help="B,C,Hs: only this shape")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    N = 64
    res = {}
    shapes = ((8, 16, 128, 1), (8, 32, 64, 1), (24, 64, 32, 3))
    if args.shape:
        shapes = (tuple(int(v) for v in args.shape.split(",")) + (1,),)
    for (B, C, Hs, G) in shapes:
        L = Hs * Hs
        g = torch.Generator().manual_seed(C)
        r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev)
        x, dy = r(B, C, L), r(B, C, Hs, Hs)
        lw, lb = torch.ones(G * C, device=dev), torch.zeros(G * C, device=dev)
        w_bcdt, w_dw = r(G * 3 * N, C, k=1 / C ** 0.5), r(G * 3 * N, 9, k=0.4)
        w_hz, w_out, D = r(G * 2 * C, C, k=1 / C ** 0.5), r(G * C, C, k=1 / C ** 0.5), torch.ones(G, device=dev)
        y, h = torch.empty(B, C, Hs, Hs, device=dev), torch.empty(B, C, N, device=dev)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev)
        xn, stats = torch.empty_like(x), torch.empty(B, L, 2, device=dev)
        wpk = torch.empty(lib.kmu_hsmssd_pack_elems(C, G), device=dev, dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.kmu_hsmssd_pack_x3(w_bcdt.data_ptr(), w_dw.data_ptr(), wpk.data_ptr(), C, G, st), "pack")
        nb = lib.kmu_mixer_fwd_ws_bytes(B, C, N, Hs)
        ws = torch.empty(nb // 4 + 1, device=dev)
        tk = torch.zeros(B, device=dev, dtype=torch.int32)
        for stage in (0, 1):
            _lib.check(lib.kmu_mixer_fwd_stage(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1e-5, w_dw.data_ptr(), w_hz.data_ptr(), w_out.data_ptr(),
                                               D.data_ptr(), wpk.data_ptr(), y.data_ptr(), h.data_ptr(), state.data_ptr(), xn.data_ptr(),
                                               stats.data_ptr(), ws.data_ptr(), nb, tk.data_ptr(), B, C, N, Hs, stage, G, st), "mixer fwd")
        P, Gp, Pc = lib.kmu_hsmssd_bwd_partials_x3(B, C, Hs), lib.kmu_hsmssd_gate_partials(B), lib.kmu_mixer_bwd_partials(B, C)
        mk = lambda *s: torch.empty(*s, device=dev)
        dx, p_bcdt, p_dw, p_hz, p_out, p_D = mk(B, C, L), mk(P, 3 * N, C), mk(P, 3 * N, 9), mk(Gp, 2 * C, C), mk(Gp, C, C), mk(Gp)
        pc_W, pc_dw = mk(Pc, N, C), mk(Pc, N, 9)
        nbb = lib.kmu_mixer_bwd_ws_bytes(B, C, N, Hs)
        wsb = torch.empty(nbb // 4 + 1, device=dev)
        nbo = lib.kmu_hsmssd_bwd_ws_bytes_x3_g(B, C, N, Hs, G)
        wso = torch.empty(nbo // 4 + 1, device=dev)

        def new(stage):
            _lib.check(lib.kmu_mixer_bwd_stage(xn.data_ptr(), dy.data_ptr(), None, w_bcdt.data_ptr(), w_dw.data_ptr(), w_hz.data_ptr(), w_out.data_ptr(),
                                               D.data_ptr(), state.data_ptr(), dx.data_ptr(), p_bcdt.data_ptr(), p_dw.data_ptr(), p_hz.data_ptr(),
                                               p_out.data_ptr(), p_D.data_ptr(), pc_W.data_ptr(), pc_dw.data_ptr(), wsb.data_ptr(), nbb, B, C, N, Hs,
                                               stage, G, wpk.data_ptr(), st), "mixer bwd")

        def old(stage):
            _lib.check(lib.kmu_hsmssd_bwd_stage_x3_pk(xn.data_ptr(), dy.data_ptr(), None, w_bcdt.data_ptr(), w_dw.data_ptr(), w_hz.data_ptr(),
                                                      w_out.data_ptr(), D.data_ptr(), state.data_ptr(), dx.data_ptr(), p_bcdt.data_ptr(),
                                                      p_dw.data_ptr(), p_hz.data_ptr(), p_out.data_ptr(), p_D.data_ptr(), wso.data_ptr(), nbo, B, C, N,
                                                      Hs, stage, G, wpk.data_ptr(), st), "old bwd")
        row = {}
        for i, nm in enumerate(("corr", "crows", "gate", "passB")):
            row["new_" + nm] = timed(lambda: new(i), args.iters)
        row["new_all"] = timed(lambda: [new(i) for i in range(4)], args.iters)
        if C <= 32:
            for mode, nm in ((0, "passB_fp32"), (1, "passB_x3")):
                lib.kmu_mixer_debug_passb(mode)
                row["new_" + nm] = timed(lambda: new(3), args.iters)
            lib.kmu_mixer_debug_passb(-1)
        for i, nm in enumerate(("passA", "gate", "passB")):
            row["old_" + nm] = timed(lambda: old(i), args.iters)
        row["old_all"] = timed(lambda: [old(i) for i in range(3)], args.iters)
        res["%d,%d,%d" % (B, C, Hs)] = {k: round(v, 2) for k, v in row.items()}
        print((B, C, Hs), res["%d,%d,%d" % (B, C, Hs)], flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
