#!/usr/bin/env python3
"""bench.py -- training frames/s of KM_UNetV3 (SH) on synthetic [B,T,1,128,128] sequences.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = the reference loop body (train_shanghai.py:163-181): slice 5 input / T-5 target frames,
forward, loss, backward, gradient all-reduce (RCCL, N > 1), AdamW step.  Workload = BASELINE.json
configs[1]: KM_UNetV3_SH, B=8 per GPU, T=10 (=> num_classes 5), 128x128; weak scaling (global batch
8*N).  Storage and accumulation are fp32 end to end; the KxK contractions form their products on the bf16
matrix core from split-bf16 operands (~16 significant bits: >= the reference's fp16 autocast; the CPU fp32
path is the parity oracle, see DTYPE below); data is synthetic (torch.rand, seed 0), weights are random-init
of that architecture.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant hand-written kernel of the step, timed per launch with HIP events on the
                  launch stream during extra instrumented steps of the same workload;
  north_star   -- BASELINE.json's "KANConv2D + SSM fused forward at B=8": the K1 + K2 forward launches of one step (LayerNorm1D
                  prologue and the once-per-step weight packs included) as ONE dependent chain on the model's own weights, captured
                  in a hipGraph and replayed: device time per replay against SURVEY 8(d)'s algorithmic bytes;
  cpu_baseline -- the CPU oracle ("port" of the reference op sequence) timed on this host's cores
                  at the headline configuration (B=8, same T/H/W; 1 warm-up + 2 timed steps), N=1 only.
The printed line stays under 2 KB (the driver keeps only a short tail of stdout); the per-entry-point table of the
instrumented steps (ms per step, GB/s, TFLOP/s for ~200 (kernel, shape) pairs) goes to --kernels-out
(default gpurun_out/bench_kernels.json) and its top rows to stderr.
"""
import argparse
import json
import os
import re
import sys
import time
import warnings

os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")       # km-unet_amd/__init__.py: numerics policy
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")    # km-unet_amd/__init__.py: hipGraph replay policy

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA (v_mfma_f32_16x16x4_f32) dense peak


# what is stored and what is multiplied: every tensor in HBM is fp32 and every accumulation is fp32; the K x K contractions (K1, K2
# forward / pass A, the plain 3x3 / 5x5 / 7x7 convs) form their products on the bf16 matrix core from operands split into two bf16
# (hi*hi + hi*lo + lo*hi, ~16 significant bits); the 1x1 contractions and K2's pass B use the exact-fp32 MFMA
DTYPE = "f32 storage + accumulate; split-bf16 (bf16x3) products in the KxK contractions"

RIDGE = PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)   # FLOP/B at which fp32 MFMA and HBM roofs cross (19.7)


def pmc_traffic(name, shape):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r04b_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes), measured at the bench shape only; None for
    kernels / shapes without a PMC pass."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "r04b_pmc_traffic.json")))["kernels"]
    except Exception:
        return None
    key = {("mixer_bwd_passB", (8, 16, 128)): "hsm_bwd_passB<16>", ("mixer_bwd_corr", (8, 16, 128)): "hsm_bwd_corr<16>",
           ("hsmssd_bwd_passB", (8, 32, 64)): "hsm_bwd_passB<32>", ("hsmssd_bwd_passA_x3", (8, 32, 64)): "hsm_bwd_passA_x3<32>",
           ("hsmssd_fwd_pass1_v2", (8, 16, 128)): "hsm2_fwd_pass1<16, 4, 8, true>", ("hsmssd_fwd_pass2_v2", (8, 16, 128)): "hsm2_fwd_pass2<16>",
           ("hsmssd_fwd_pass1_v2", (8, 32, 64)): "hsm2_fwd_pass1<32, 1, 8, false>", ("hsmssd_fwd_pass2_v2", (8, 32, 64)): "hsm2_fwd_pass2<32>",
           ("kan_conv2d_fwd_x3", (8, 16, 16, 128, 128)): "conv3x3_x3_fwd_kernel<0, 3, 8, 32, 4, 1, 1>",
           ("kan_conv2d_bwd_input_x3", (8, 16, 16, 128, 128)): "kan_dgrad_x3_kernel<4, 32>",
           ("pwconv_fwd", (8, 16, 64, 16384)): "pw_gemm_kernel<4>"}.get((name, tuple(shape)))
    return table[key]["hbm_bytes"] if key in table else None


def kernel_model(name, shape):
    """Algorithmic work of one launch (DESIGN.md 'Kernels'): (bound, flops, bytes).  The binding roof is chosen by
    arithmetic intensity against the fp32-MFMA / HBM ridge; gather / streaming kernels are HBM-bound."""
    bound, flops, byts = _kernel_model(name, shape)
    if flops and byts:
        bound = "mfma" if flops / byts > RIDGE else "hbm"
    return bound, flops, byts


def _kernel_model(name, shape):
    # entry-point suffixes name the arithmetic / calling form, not the work: _x3 (split-bf16 matrix core), _v2 (round-4 forward),
    # _g (grouped: the shape's B already counts samples x groups), _pk (pack handed in)
    base, again = name, True
    while again:
        again = False
        for suf in ("_pk", "_g", "_x3", "_v2"):
            if base.endswith(suf):
                base, again = base[:-len(suf)], True
    if base.startswith("kan_conv2d"):
        B, Cin, Cout, H, W = shape
        flops = 2.0 * B * H * W * (81 * Cin) * Cout           # implicit GEMM, K = 9 taps x 9 basis x Cin
        byts = 4.0 * B * H * W * (Cin + Cout) + 4.0 * 81 * Cin * Cout
        if base.endswith("bwd_input"):
            byts = 4.0 * B * H * W * (2 * Cin + Cout) + 4.0 * 81 * Cin * Cout
        if base.endswith("bwd_weights"):
            byts = 4.0 * B * H * W * (Cin + Cout) + 4.0 * 81 * Cin * Cout
        return "mfma", flops, byts
    if base.startswith("mixer_bwd"):       # csrc/hsmssd_bwdc.inc: the K2 backward with the C rows as a per-sample dense convolution
        B, C, Hs = shape
        L, N = Hs * Hs, 64
        TT = -(-Hs // 8) * -(-Hs // (32 if C <= 32 else 16))
        slabs = 4.0 * B * min(TT, max(1, 1024 // B)) * 9 * C * C          # G partials: one [C][9][C] slab per workgroup
        if base.endswith("corr"):
            return "hbm", B * L * 2.0 * 9 * C * C, 4.0 * B * C * L * 2 + slabs
        if base.endswith("crows"):
            return "hbm", B * 4.0 * 9 * C * C * N, slabs
        proj, dw, mix = 2.0 * N * C, 2.0 * 9 * N, 2.0 * C * N
        # pass B on the {B, dt} rows (projection recompute, dx, dW; stencil forward / transposed / weight gradient; dAB and its dx) + the
        # dense transposed convolution of the C rows
        return "hbm", B * L * (2 * 3 * proj + 2 * 3 * dw + 2 * mix + 2.0 * 9 * C * C), 4.0 * B * C * L * 3
    if base.startswith("hsmssd"):
        B, C, Hs = shape
        L, N = Hs * Hs, 64
        proj, dw, mix = 2.0 * N * C, 2.0 * 9 * N, 2.0 * C * N       # per token, per group of N rows of BCdt
        v2 = "_v2" in name
        # tensors = algorithmic [B,C,L] reads + writes.  v2 pass 2 is the per-sample dense 3x3 (9 C^2 MACs per token) and also writes the
        # normalised x for the backward pass
        per_token = {"fwd_pass1": (2 * proj + 2 * dw + mix, 1), "fwd_pass2": ((2.0 * 9 * C * C, 3) if v2 else (proj + dw + mix, 2)),
                     "bwd_passA": (proj + dw + mix, 2), "bwd_passB": (3 * proj * 3 + 6 * dw + 4 * mix, 3)}
        for key, (fl, tensors) in per_token.items():
            if base.endswith(key):
                return "hbm", B * L * fl, 4.0 * B * C * L * tensors
        # gate stages: the tile partials in, the [C, 64] state out (8x8 token tiles of 64 states x C channels, + m / s rows)
        return "hbm", 0.0, 4.0 * B * (L / 64.0) * (C * N + 2 * N) / 4.0 + 4.0 * B * C * N * 8
    if name.startswith("dysample"):
        B, C, H, W = shape
        byts = 4.0 * B * C * H * W * 5 + 4.0 * B * 32 * H * W
        if name.endswith("bwd"):
            byts *= 2
        return "hbm", 12.0 * B * C * 4 * H * W, byts
    # ---- streaming glue kernels: algorithmic bytes = every operand read / written once, fp32 ----
    t = 4.0                                                       # bytes per element
    if name.startswith("pwconv"):
        B, Ci, Co, P = shape
        flops = 2.0 * B * P * Ci * Co
        extra = B * P * Ci if name.endswith("bwd_input") else 0     # GELU'(x_pre) operand (upper bound: only with gelu_in)
        return "hbm", flops, t * (B * P * (Ci + Co) + Ci * Co) + 0 * extra
    if name.startswith("tail_ffn"):   # csrc/ffn_fused.hip: fwd reads nrm, x and writes out; bwd reads nrm, g and writes dn
        B, C, P = shape
        prod = 2.0 * B * P * C * 4 * C
        return "hbm", (2 if name.endswith("fwd") else 5) * prod, t * B * C * P * 3
    if name.startswith("ffn_"):       # csrc/ffn_fused.hip: [B,C,P] tensors read / written once per pass, 4C hidden channels recomputed
        B, C, P = shape
        prod = 2.0 * B * P * C * 4 * C                               # one C x 4C product over all pixels
        tensors, nprod = {"ffn_fwd_stats": (1, 1), "ffn_fwd_main": (2, 2), "ffn_fwd_apply": (3, 0), "ffn_bwd_red": (3, 0),
                          "ffn_bwd_mid": (3, 3), "ffn_bwd_in": (4, 4)}[name]
        return "hbm", nprod * prod, t * B * C * P * tensors
    if name.startswith("bn_blend_fwd_pre"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * 3                      # t once (the conv left the statistics), x, out
    if name.startswith("bn_blend"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * (4 if name.endswith("fwd") else 7)   # fwd: t twice (stats, apply), x, out
    if name.startswith("triple_norm"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * (3 if name.endswith("fwd") else 6)   # fwd: x twice, y; bwd: x, dy twice, addend, dx
    if name.startswith("mean_rows"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW
    if name.startswith("lca_"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * (2 if name.endswith("fwd") else 3)
    if name.startswith(("dagem_fwd", "dagem_bwd")):       # csrc/dagem_fused.hip: [B,C,HW]-sized tensors read / written per stage
        B, C, H, W = shape
        tensors = {"dagem_fwd0": 4, "dagem_fwd1": 5, "dagem_fwd2": 4, "dagem_fwd3": 2, "dagem_bwd0": 2, "dagem_bwd1": 6, "dagem_bwd2": 8,
                   "dagem_bwd3": 9, "dagem_bwd4": 7}[name]
        return "hbm", 2.0 * B * H * W * C * C * 3, t * B * C * H * W * tensors
    if name.startswith("dagem_edges"):
        B, C, H, W = shape
        return "hbm", 0.0, t * B * C * H * W * (5 if name.endswith("fwd") else 6)
    if name.startswith("bn_blend_bwd_partials"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * 3
    if name.startswith("qkv_dw_scaled"):       # qkv [B,3C,H,W] read once (+ dy), out / dqkv written once
        B, C, H, W = shape
        return "hbm", 60.0 * B * C * H * W, t * B * C * H * W * (4 if name.endswith("fwd") else 7)
    if name.startswith("dwconv3x3_bn_bwd"):
        B, C, H, W = shape
        return "hbm", 24.0 * B * C * H * W, t * B * C * H * W * 3
    if name.startswith("dwconv3x3"):
        B, C, H, W = shape[:4] if len(shape) >= 4 else (shape[0], shape[1], 1, 1)
        return "hbm", 18.0 * B * C * H * W, t * B * C * H * W * 2
    if name.startswith(("layernorm1d", "group_norm")):
        n = 1
        for d in shape:
            n *= d
        return "hbm", 0.0, t * n * (2 if name.endswith("fwd") else 3)
    if name.startswith("qkv_gate"):
        B, C, HW = shape
        return "hbm", 0.0, t * B * C * HW * (4 if name.endswith("fwd") else 7)
    if name.startswith("mix3"):
        B, n = shape
        return "hbm", 0.0, t * B * n * (5 if name.endswith("fwd") else 7)
    if name.startswith("shift3"):
        B, C, H, W = shape
        return "hbm", 0.0, t * B * C * H * W * 4
    if name.startswith("iwp_front"):
        B, C, H, W = shape
        return "hbm", 0.0, t * B * H * W * (C + (C + 1) / 4.0)
    if name.startswith("gauss11"):
        N, H, W = shape
        return "hbm", 2.0 * 22 * N * H * W, t * N * H * W * 2
    m = re.match(r"conv(\d)x\d_(fwd|dgrad|bwd_weight)", base)
    if m:                                         # plain K x K convolutions on the bf16 matrix core: [B, Cin, Cout, H, W]
        B, Cin, Cout, H, W = shape
        K = int(m.group(1))
        flops = 2.0 * B * H * W * K * K * Cin * Cout
        return "mfma", flops, t * (B * H * W * (Cin + Cout) + K * K * Cin * Cout)
    if base.startswith("add_n"):
        n, numel = shape
        return "hbm", 0.0, t * numel * (n + 1)
    if base.startswith("relu_mask"):
        return "hbm", 0.0, t * shape[0] * 3
    if base.startswith(("colsum_multi", "bias_sum_multi", "copy_multi")):     # shape = element counts of the partial arrays / jobs
        return "hbm", 0.0, t * max(1, sum(shape)) * (2 if base.startswith("copy") else 1)
    if base.startswith("deform_sample"):
        B, C, H, W = shape
        return "hbm", 0.0, t * B * H * W * (C + 18 + 9 * C) * (2 if base.endswith("bwd") else 1)
    if base.startswith("gate_mlp"):               # pooled vectors through a two-layer MLP: a few KB
        B, I, Hd, O = shape
        return "hbm", 2.0 * B * (I * Hd + Hd * O), t * (B * (I + O) + I * Hd + Hd * O)
    if base.startswith(("conv_pack_multi", "hsm_pack_multi")):                # weight re-layouts: all packed weights once (~2 x 1.7 M params)
        return "hbm", 0.0, t * 1.7e6 * 2
    if base.startswith("hybrid_loss"):
        N, H, W = shape
        return "hbm", 0.0, t * N * H * W * {"stats": 2, "stack": 7, "combine": 5, "grad_maps": 8, "grad_input": 8}.get(base.split("hybrid_loss_")[1], 4)
    if base.startswith("resize_bilinear"):
        B, C, Hi, Wi, Ho, Wo = shape
        return "hbm", 0.0, t * B * C * (Hi * Wi + Ho * Wo)
    return "hbm", 0.0, 0.0


def k1k2_forward_chain(model, data, replays=20):
    """K1 + K2 forward of one training step as ONE dependent chain: the live KANConv2d sites and every mixer (LayerNorm1D + HSMSSD; the
    levels the step runs stacked are stacked here too) on random inputs of the shapes the model gives them, with the model's weights, in
    train form (the normalised x and the LayerNorm statistics are written for the backward pass), behind the once-per-step weight packs.
    Captured in a hipGraph and replayed: (us per replay, kernel launches per replay)."""
    import km_unet_amd.model as M
    import km_unet_amd.nn as NN
    from km_unet_amd import grouped, ops
    from km_unet_amd.train import split_frames
    shapes = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp: shapes.__setitem__(mod, tuple(inp[0].shape)))
             for m in model.modules() if isinstance(m, (M.EnhancedViMBlock, NN.KANConv2d))]
    with torch.no_grad():
        model(split_frames(data)[0])
    for h in hooks:
        h.remove()
    jobs, launches = [], [0]
    gen = torch.Generator(device=data.device).manual_seed(5)

    def mixer_args(e):
        mx = e.mixer
        return (e.norm.weight, e.norm.bias, e.norm.eps, mx.BCdt_proj.conv.weight, mx.dw.conv.weight, mx.hz_proj.conv.weight, mx.out_proj.conv.weight, mx.A, mx.D)

    for mod, shp in shapes.items():
        x = torch.randn(shp, device=data.device, generator=gen).requires_grad_(True)
        if isinstance(mod, NN.KANConv2d):
            k = mod.kanlayer
            jobs.append(lambda x=x, k=k: ops.kan_conv2d(x, k.grid, k.base_weight, k.spline_weight, k.spline_scaler))
            launches[0] += 1
            continue
        blocks = (mod.height_block, mod.width_block, mod.channel_block)
        Bq, C, Hh, Ww = shp
        if M._GROUPED_BRANCHES and C >= M._GROUPED_MIN_C and grouped.supported(blocks, x):
            ev = [b.vit_mamba for b in blocks]
            xs = torch.randn(Bq * grouped.G, C, Hh * Ww, device=data.device, generator=gen).requires_grad_(True)

            def gjob(ev=ev, xs=xs):
                sp = grouped.stack_params
                mx = [e.mixer for e in ev]
                return grouped.MixerGFn.apply(xs, sp([e.norm.weight for e in ev]), sp([e.norm.bias for e in ev]), ev[0].norm.eps,
                                              sp([m.BCdt_proj.conv.weight for m in mx]), sp([m.dw.conv.weight for m in mx]),
                                              sp([m.hz_proj.conv.weight for m in mx]), sp([m.out_proj.conv.weight for m in mx]),
                                              sp([m.A for m in mx]), sp([m.D for m in mx]))
            jobs.append(gjob)
            launches[0] += 2
        else:
            for b in blocks:
                e = b.vit_mamba
                xs = torch.randn(Bq, C, Hh * Ww, device=data.device, generator=gen).requires_grad_(True)
                jobs.append(lambda e=e, xs=xs: ops.mixer_ln(xs, *mixer_args(e), alias=True))
                launches[0] += 2

    def chain():
        with ops.pack_scope():
            ops.prepack()
            return [j() for j in jobs]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            chain()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = chain()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(replays):
        g.replay()
    e_.record()
    torch.cuda.synchronize()
    del keep
    npacks = sum(1 for t in (ops._PACKS.plan or []) if t is not None)
    return s_.elapsed_time(e_) / replays * 1e3, launches[0] + npacks


def cpu_baseline(T, H, B=8, steps=2, loss="hybrid"):
    """CPU oracle (oracle/model.py = the reference's op sequence) fwd + loss + bwd + AdamW at the headline batch."""
    from oracle.loss import hybrid_loss
    from oracle.model import KM_UNetV3 as Oracle
    crit = hybrid_loss if loss == "hybrid" else torch.nn.functional.mse_loss
    torch.manual_seed(0)
    # one GPU's host share on the box is 16 cores; more threads only add oversubscription on these
    # small tensors (measured: 128 threads = 14.6 s/step, slower than 8 threads in the build container)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m = Oracle(num_classes=T - 5).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.05)
    data = torch.rand(B, T, 1, H, H)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        d = data.squeeze(2)
        opt.zero_grad()
        lv = crit(m(d[:, :5]), d[:, 5:])
        lv.backward()
        opt.step()
        if i:
            times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2]
    return {"value": B * T / t, "unit": "frames/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": "oracle KM_UNetV3 train step (fwd+%s loss+bwd+AdamW) at B=%d,T=%d,%dx%d fp32, median of %d steps after 1 warm-up, %.2f s/step"
                      % (loss, B, T, H, H, steps, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss", choices=("hybrid", "mse"), default="hybrid",
                    help="hybrid = the reference's HybridLoss (train_shanghai.py:298-325), the default; mse = plain MSE")
    ap.add_argument("--no-graph", action="store_true", help="run eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--kernels-out", default=os.path.join(ROOT, "gpurun_out", "bench_kernels.json"),
                    help="where the per-entry-point table of the instrumented steps is written (not on stdout)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (children of a parent that never touched the
        # GPU -- no exec from a HIP-initialised process) and relay rank 0's line
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29531"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # KMU_DIST_BACKEND=gloo lets the N>1 code path be rehearsed with several ranks sharing one GPU
    # (RCCL refuses two ranks on one device); the real runs use "nccl" (= RCCL over xGMI).
    backend = os.environ.get("KMU_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # KMU_FORCE_DIST=1: take the distributed code path (process group, two-graph step, eager all-reduce between the graphs)
    # with a single rank too, so that the RCCL path runs on a one-GPU box (tests/test_gpu_model.py)
    force_dist = os.environ.get("KMU_FORCE_DIST", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import km_unet_amd
    from km_unet_amd import ops
    from km_unet_amd.train import GraphedTrainStep, TrainStep

    B, T, H = args.batch, args.frames, args.size
    torch.manual_seed(0)
    model = km_unet_amd.KM_UNetV3(num_classes=T - 5).to(dev).train()
    torch.manual_seed(1234 + rank)            # per-rank data shard and DropPath stream
    data = torch.rand(B, T, 1, H, H, device=dev)
    eager = TrainStep(model, data, capturable=not args.no_graph, loss=args.loss, force_collective=force_dist)
    step = eager if args.no_graph else GraphedTrainStep(eager, data)

    def sync():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    loss = step(data)                        # one extra untimed step: reference point of the sanity check below
    loss_ref = loss.item()
    for _ in range(args.warmup):
        loss = step(data)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(data)
    host_dt = time.perf_counter() - t0      # host time to ISSUE the steps (graph launches return before the device is done)
    sync()
    dt = time.perf_counter() - t0
    # one step from an idle device: how long the host is held by the launch itself (a graph launch that returned only when the
    # device was done would make the step host-bound)
    lone = []
    for _ in range(5):
        sync()
        ta = time.perf_counter()
        step(data)
        tb = time.perf_counter()
        torch.cuda.synchronize()
        lone.append((tb - ta, time.perf_counter() - ta))
    lone.sort()
    host_launch_ms, lone_step_ms = lone[2][0] * 1e3, sorted(t[1] for t in lone)[2] * 1e3
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    final_loss = loss.item()
    # the same batch is fitted over and over, so the loss can only drift down; a jump means the replayed step no
    # longer computes the step (see km-unet_amd/__init__.py on DEBUG_CLR_GRAPH_PACKET_CAPTURE) -> no number at all
    if not (final_loss == final_loss and final_loss <= 1.5 * loss_ref + 1e-3):
        raise RuntimeError("bench: loss went %.5g -> %.5g over the timed steps; the measured step is not a valid "
                           "training step" % (loss_ref, final_loss))

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {"metric": "train frames/sec (BxT) KM-UNetV3_SH 128x128 T=10", "value": world * B * T / (dt / args.steps),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
               "config": {"workload": "KM_UNetV3_SH(num_classes=%d) train step (fwd + %s loss + bwd + grad all-reduce + AdamW), "
                                      "B=%d per GPU, T=%d, %dx%d -- BASELINE.json configs[1]" % (T - 5, args.loss, B, T, H, H),
                          "global_batch": world * B, "frames_per_sample": T, "parallelism": "dp%d" % world},
               "loss_first": loss_ref, "loss": final_loss, "launch_mode": "eager" if args.no_graph else "hipGraph replay",
               "host_issue_ms_per_step": round(host_dt / args.steps * 1e3, 3), "host_launch_ms": round(host_launch_ms, 3),
               "lone_step_ms": round(lone_step_ms, 3),
               "collective": ("%s all-reduce (average) of %d floats per step over %d rank(s), world %d" % (
                   backend, eager.dp.bucket.numel(), dist.get_world_size(), world)) if eager.dp.collective else "none (1 rank)"}

    # ---- roofline leg: extra instrumented steps, HIP events around every C-ABI launch ----------
    # (every rank runs them -- a step contains the gradient all-reduce -- only rank 0 records and reports)
    nprof = 3
    if rank == 0:
        ops.profile_begin()
    for _ in range(nprof):
        eager(data)                      # eager: HIP events bracket each launch on the launch stream
    if rank == 0:
        prof = ops.profile_end()
        table = {}
        for (name, shape), ms_list in prof.items():
            bound, flops, byts = kernel_model(name, shape)
            per = sum(ms_list) / len(ms_list)
            table["%s%s" % (name, list(shape))] = {
                "launches_per_step": len(ms_list) / nprof, "avg_ms": per, "ms_per_step": sum(ms_list) / nprof,
                "bound": bound, "GB/s": byts / per / 1e6 if per else 0.0, "TFLOP/s": flops / per / 1e9 if per else 0.0}
        dom = max(table, key=lambda k: table[k]["ms_per_step"])
        d = table[dom]
        (name, shape) = next(k for k in prof if "%s%s" % (k[0], list(k[1])) == dom)
        bound, flops, byts = kernel_model(name, shape)
        if bound == "mfma":
            # `frac` is against the roof of the instruction the kernel issues (exact-fp32 MFMA, 157.3 TF); every other contraction of
            # the step runs split-bf16 on the bf16 matrix core (2.5 PF / 3 products = 833 TF effective): that fraction beside it
            roof = {"bound": "mfma", "achieved": d["TFLOP/s"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": d["TFLOP/s"] / PEAK_F32_MFMA_TFLOPS, "frac_bf16x3_roof": d["TFLOP/s"] / (2500.0 / 3.0)}
        else:
            roof = {"bound": "hbm", "achieved": d["GB/s"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": d["GB/s"] / PEAK_HBM_GBS}
        roof.update({"traffic": pmc_traffic(name, shape), "kernel": dom, "avg_launch_ms": d["avg_ms"], "algorithmic_flops": flops,
                     "algorithmic_bytes": byts, "hbm_GBps_on_algorithmic_bytes": d["GB/s"]})
        out["roofline"] = {k: (float("%.5g" % v) if isinstance(v, float) else v) for k, v in roof.items()}
        out["hip_kernels_ms_per_step"] = round(sum(v["ms_per_step"] for v in table.values()), 4)
        # BASELINE.json north_star: "the KANConv2D+SSM fused forward at B=8": K1 + K2 forward (LayerNorm1D prologue and weight packs included),
        # (a) as one dependent chain replayed from a hipGraph (the figure), (b) the HIP-event sum of the instrumented steps' launches
        pre = ("kan_conv2d_fwd", "hsmssd_fwd", "conv_pack_multi", "hsm_pack_multi", "layernorm1d_fwd")
        ns = [v for k, v in table.items() if k.startswith(pre)]
        ev_us = 1e3 * sum(v["ms_per_step"] for v in ns)
        ns_bytes = 24.2e6 * B * (H / 128.0) ** 2
        chain_us, chain_launches = k1k2_forward_chain(model, data)
        out["north_star"] = {"k1k2_fwd_us": round(chain_us, 1), "launches": chain_launches, "algorithmic_bytes": ns_bytes,
                             "hbm_frac": float("%.4g" % (ns_bytes / (chain_us * 1e-6) / (PEAK_HBM_GBS * 1e9))),
                             "method": "hipGraph replay of the K1+K2 forward chain (LayerNorm + packs included)",
                             "event_sum_us": round(ev_us, 1), "event_launches": int(round(sum(v["launches_per_step"] for v in ns)))}
        kernels = {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in
                   sorted(table.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        try:
            os.makedirs(os.path.dirname(args.kernels_out), exist_ok=True)
            with open(args.kernels_out, "w") as f:
                json.dump({"config": out["config"], "ms_per_step": ms, "kernels": kernels}, f, indent=1)
            out["kernels_table"] = os.path.relpath(args.kernels_out, ROOT)
        except OSError as e:                          # read-only checkout: the table is a convenience, not the contract
            out["kernels_table"] = "not written (%s)" % e
        for k, v in list(kernels.items())[:12]:
            print("bench: %-52s %6.3f ms/step  %5.1f launches  %s" % (k, v["ms_per_step"], v["launches_per_step"], v["bound"]),
                  file=sys.stderr)

    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, H, B=B, loss=args.loss)
        for k in ("value", "ms_per_step", "loss_first", "loss"):
            out[k] = float("%.6g" % out[k])
        line = json.dumps(out)
        assert len(line) < 2048, "bench line too long for the driver's stdout tail: %d" % len(line)
        sys.stderr.flush()
        print(line, flush=True)                       # the contract line: last thing on stdout
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
