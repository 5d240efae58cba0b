"""The three direction branches of EnhancedViMBlock as ONE stacked pass (KM_UNetV3_SH.py:97-151, :154-263).

height / width / channel DirectionViM differ only in their first projection; afterwards each runs the same EfficientViMBlock
(efficient_vim_init.py:64-97) and DirectionAttention (:215-263) on a tensor of the same shape with its own weights.  The
reference runs them one after the other; round 2 first forked them onto three HIP streams, which overlaps little on this
runtime (tools/time_block.py: 1647 us for the block at C = 64 against 3 x 600 serial).  Here the branch index becomes part of the
channel axis -- x [B, 3C, H, W], parameters stacked [3, ...] -- so every layer is one launch with three times the workgroups:
these launches are latency-bound (2-8 MB tensors on 256 CUs), three times the work costs 1.7x (C = 64) .. 2.4x (C = 16) the time
of one branch instead of 3x.  Per-channel layers (depthwise conv, BatchNorm blend, qkv gate, pooling) run unchanged on 3C
channels; layers that mix channels have grouped entry points (include/kmunet_hip.h, "Grouped variants").

The stacked parameters are views, not copies, whenever the three modules' tensors lie back to back in memory: DataParallel orders
the flat parameter buffer that way (dp.branch_adjacent_order); a bare model pays one multi-tensor copy per stacked parameter.
"""
import torch

from . import _lib, ops
from .ops import _call, _f32c, _leaf, _ptr, _stream, _wgrad, colsum

G = 3


# ------------------------------------------------------------------------------------------ parameter stacking
def _adjacent(ts):
    t0 = ts[0]
    if not (t0.is_cuda and t0.dtype == torch.float32 and t0.is_contiguous()):
        return False
    n, st = t0.numel() * 4, t0.untyped_storage().data_ptr()
    return all(t.is_contiguous() and t.dtype == t0.dtype and t.shape == t0.shape and t.untyped_storage().data_ptr() == st and
               t.data_ptr() == t0.data_ptr() + i * n for i, t in enumerate(ts))


class StackParamsFn(torch.autograd.Function):
    """(p_0, .., p_{G-1}) of one shape [d0, ...] -> [G*d0, ...].  A zero-copy view of the span when the tensors are adjacent in one
    storage, else a copy.  Backward: each parameter gets its slice of the stacked gradient (views; nothing is read, so a consumer
    may hand the gradient over unfilled -- ops._leaf accepts this node)."""

    @staticmethod
    def forward(ctx, *ps):
        p0 = ps[0]
        ctx.shape = tuple(p0.shape)
        lead = (len(ps) * p0.shape[0],) + tuple(p0.shape[1:]) if p0.dim() else (len(ps),)
        if _adjacent(ps):
            return torch.as_strided(p0, lead, torch.empty(lead, device="meta").stride(), p0.storage_offset())
        return torch.cat([p.reshape((1,) if p.dim() == 0 else p.shape) for p in ps]).view(lead)

    @staticmethod
    def backward(ctx, g):
        n = g.shape[0] // G if len(ctx.shape) else 1
        return tuple(g[i * n:(i + 1) * n].view(ctx.shape) for i in range(G))


def stack_params(ps):
    return StackParamsFn.apply(*ps)


def stacked_buffer(bufs):
    """One tensor aliasing G same-shaped module buffers (BatchNorm running statistics).  When they are not adjacent yet they are
    moved into one new storage and the modules' buffers re-pointed at its slices (once; buffers are not part of the flat
    parameter buffer, so nothing else aliases them)."""
    if not _adjacent(bufs):
        st = torch.cat([b.detach().reshape(-1) for b in bufs])
        n = bufs[0].numel()
        for i, b in enumerate(bufs):
            b.data = st[i * n:(i + 1) * n].view(b.shape)
    b0 = bufs[0]
    lead = (len(bufs) * b0.numel(),)
    return torch.as_strided(b0, lead, (1,), b0.storage_offset())


# ------------------------------------------------------------------------------------------ LayerNorm1D, grouped
class LayerNorm1dGFn(torch.autograd.Function):
    """LayerNorm1D over C of x [B*G, C, L] (sample b: group b % G), weight / bias [G*C] (vim_utils_init.py:50-59)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        x = _f32c(x, "x")
        w, b = _f32c(weight, "weight").reshape(-1), _f32c(bias, "bias").reshape(-1)
        B, C, L = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B, L, 2, device=x.device, dtype=torch.float32)
        _lib.check(_call(("layernorm1d_fwd_g", (B, C, L)), lib.kmu_layernorm1d_fwd_g, _ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(stats), B, C,
                         L, float(eps), G, _stream()), "kmu_layernorm1d_fwd_g")
        ctx.save_for_backward(x, w, stats)
        ctx.wshape = weight.shape
        ctx.defer_wgrad = _leaf(weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, stats = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C, L = x.shape
        rows = lib.kmu_layernorm1d_partials(B, C, L)
        dx = torch.empty_like(x)
        dwp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        dbp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        _lib.check(_call(("layernorm1d_bwd_g", (B, C, L)), lib.kmu_layernorm1d_bwd_g, _ptr(x), _ptr(w), _ptr(stats), _ptr(dy), _ptr(dx),
                         _ptr(dwp), _ptr(dbp), B, C, L, G, _stream()), "kmu_layernorm1d_bwd_g")
        dw = torch.empty(G, C, device=x.device, dtype=torch.float32)
        db = torch.empty(G, C, device=x.device, dtype=torch.float32)
        nb = rows // B                      # partial rows are (sample b, block): group of a row = b % G

        def job():
            torch.sum(dwp.view(B // G, G, nb, C), dim=(0, 2), out=dw)
            torch.sum(dbp.view(B // G, G, nb, C), dim=(0, 2), out=db)
        _wgrad(job, ctx.defer_wgrad)
        return dx, dw.view(ctx.wshape), db.view(ctx.wshape), None


# ------------------------------------------------------------------------------------------ HSMSSD, grouped
class HsmssdGFn(torch.autograd.Function):
    """y [B*G, C, Hs, Hs] = HSMSSD(x [B*G, C, L]) with G weight sets (sample b: set b % G), bf16x3 kernels
    (efficient_vim_init.py:33-61).  The hidden state h is not returned (EfficientViMBlock drops it)."""

    @staticmethod
    def forward(ctx, x, w_bcdt, w_dw, w_hz, w_out, A, D):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, L = x.shape
        Hs = int(round(L ** 0.5))
        if Hs * Hs != L:
            raise RuntimeError("HSMSSD: L=%d is not a perfect square (reference: int(math.sqrt(L)))" % L)
        N = A.shape[0] // G
        ctx.defer_wgrad = _leaf(w_bcdt, w_dw, w_hz, w_out, D)
        w_bcdt, w_dw = _f32c(w_bcdt, "BCdt_proj.weight").reshape(G * 3 * N, C), _f32c(w_dw, "dw.weight").reshape(G * 3 * N, 9)
        w_hz, w_out = _f32c(w_hz, "hz_proj.weight").reshape(G * 2 * C, C), _f32c(w_out, "out_proj.weight").reshape(G * C, C)
        D = _f32c(D, "D").reshape(G)
        dev = x.device
        y = torch.empty(B, C, Hs, Hs, device=dev, dtype=torch.float32)
        h = torch.empty(B, C, N, device=dev, dtype=torch.float32)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev, dtype=torch.float32)
        nbytes = lib.kmu_hsmssd_fwd_ws_bytes_g(B, C, N, Hs, G)
        ws = torch.empty(max(1, nbytes // 4), device=dev, dtype=torch.float32)
        st = _stream()
        for stage, nm in enumerate(("hsmssd_fwd_pass1_x3", "hsmssd_fwd_gate", "hsmssd_fwd_pass2_x3")):
            _lib.check(_call((nm + "_g", (B, C, Hs)), lib.kmu_hsmssd_fwd_stage_x3_g, _ptr(x), _ptr(w_bcdt), _ptr(w_dw), _ptr(w_hz), _ptr(w_out),
                             _ptr(D), _ptr(y), _ptr(h), _ptr(state), _ptr(ws), nbytes, B, C, N, Hs, stage, G, st), "kmu_hsmssd_fwd_stage_x3_g")
        ctx.save_for_backward(x, w_bcdt, w_dw, w_hz, w_out, D, state)
        ctx.dims = (B, C, N, Hs)
        ctx.zero_A = ops._const_zeros(A)
        ctx.shapes = (tuple(A.shape),)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_bcdt, w_dw, w_hz, w_out, D, state = ctx.saved_tensors
        B, C, N, Hs = ctx.dims
        dev = x.device
        dy = _f32c(dy, "dy")
        P = lib.kmu_hsmssd_bwd_partials_x3(B, C, Hs)
        dx = torch.empty_like(x)
        p_bcdt = torch.empty(P, 3 * N, C, device=dev, dtype=torch.float32)
        p_dw = torch.empty(P, 3 * N, 9, device=dev, dtype=torch.float32)
        Gp = lib.kmu_hsmssd_gate_partials(B)
        p_hz = torch.empty(Gp, 2 * C, C, device=dev, dtype=torch.float32)
        p_out = torch.empty(Gp, C, C, device=dev, dtype=torch.float32)
        p_D = torch.empty(Gp, device=dev, dtype=torch.float32)
        nbytes = lib.kmu_hsmssd_bwd_ws_bytes_x3_g(B, C, N, Hs, G)
        ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
        st = _stream()
        for stage, nm in enumerate(("hsmssd_bwd_passA_x3", "hsmssd_bwd_gate", "hsmssd_bwd_passB")):
            _lib.check(_call((nm + "_g", (B, C, Hs)), lib.kmu_hsmssd_bwd_stage_x3_g, _ptr(x), _ptr(dy), None, _ptr(w_bcdt), _ptr(w_dw),
                             _ptr(w_hz), _ptr(w_out), _ptr(D), _ptr(state), _ptr(dx), _ptr(p_bcdt), _ptr(p_dw), _ptr(p_hz), _ptr(p_out),
                             _ptr(p_D), _ptr(ws), nbytes, B, C, N, Hs, stage, G, st), "kmu_hsmssd_bwd_stage_x3_g")
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        d_bcdt, d_dw, d_hz, d_out, d_D = mk(G, 3 * N * C), mk(G, 3 * N * 9), mk(G, 2 * C * C), mk(G, C * C), mk(G)
        Bs, tb, gp = B // G, P // B, Gp // B    # partial rows are (sample b, tile / gate block): the group of a row is b % G

        def job():
            torch.sum(p_bcdt.view(Bs, G, tb, -1), dim=(0, 2), out=d_bcdt)
            torch.sum(p_dw.view(Bs, G, tb, -1), dim=(0, 2), out=d_dw)
            torch.sum(p_hz.view(Bs, G, gp, -1), dim=(0, 2), out=d_hz)
            torch.sum(p_out.view(Bs, G, gp, -1), dim=(0, 2), out=d_out)
            torch.sum(p_D.view(Bs, G, gp), dim=(0, 2), out=d_D)
        _wgrad(job, ctx.defer_wgrad)
        return (dx, d_bcdt.view(G * 3 * N, C, 1), d_dw.view(G * 3 * N, 1, 3, 3), d_hz.view(G * 2 * C, C, 1), d_out.view(G * C, C, 1),
                ctx.zero_A, d_D.view(G))


class MixerGFn(torch.autograd.Function):
    """(y [B*G, C, Hs, Hs], x') = HSMSSD(LayerNorm1D(x [B*G, C, L])) with G weight sets (sample b: set b % G), the two launches of
    csrc/hsmssd_v2.inc (ops.MixerFn, grouped); x' aliases x for the blend that follows: its gradient is added inside the LayerNorm
    backward kernel.  Backward = ops._mixer_backward with G weight groups + the grouped LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w_bcdt, w_dw, w_hz, w_out, A, D):
        lib = _lib.load()
        xin = x
        x = _f32c(x, "x")
        B, C, L = x.shape
        Hs = int(round(L ** 0.5))
        if Hs * Hs != L:
            raise RuntimeError("HSMSSD: L=%d is not a perfect square (reference: int(math.sqrt(L)))" % L)
        N = A.shape[0] // G
        ctx.defer_wgrad = _leaf(w_bcdt, w_dw, w_hz, w_out, D)
        ctx.defer_ln = _leaf(ln_w, ln_b)
        origs = (w_bcdt, w_dw)
        w_bcdt, w_dw = _f32c(w_bcdt, "BCdt_proj.weight").reshape(G * 3 * N, C), _f32c(w_dw, "dw.weight").reshape(G * 3 * N, 9)
        ctx.pack_ok = ops._pack_ok(origs, (w_bcdt, w_dw))
        w_hz, w_out = _f32c(w_hz, "hz_proj.weight").reshape(G * 2 * C, C), _f32c(w_out, "out_proj.weight").reshape(G * C, C)
        D = _f32c(D, "D").reshape(G)
        lw, lb = _f32c(ln_w, "norm.weight").reshape(-1), _f32c(ln_b, "norm.bias").reshape(-1)
        dev = x.device
        need_bwd = any(ctx.needs_input_grad)
        xn = torch.empty_like(x) if need_bwd else None
        stats = torch.empty(B, L, 2, device=dev, dtype=torch.float32) if need_bwd else None
        y = torch.empty(B, C, Hs, Hs, device=dev, dtype=torch.float32)
        h = torch.empty(B, C, N, device=dev, dtype=torch.float32)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev, dtype=torch.float32)
        nbytes = lib.kmu_mixer_fwd_ws_bytes(B, C, N, Hs)
        ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
        st = _stream()
        wpk = ops._hsm_pack(ctx.pack_ok, w_bcdt, w_dw, C, st, G)
        tk = ops._tickets(dev, B)
        for stage, nm in enumerate(("hsmssd_fwd_pass1_v2", "hsmssd_fwd_pass2_v2")):
            _lib.check(_call((nm + "_g", (B, C, Hs)), lib.kmu_mixer_fwd_stage, _ptr(x), _ptr(lw), _ptr(lb), float(eps), _ptr(w_dw), _ptr(w_hz),
                             _ptr(w_out), _ptr(D), _ptr(wpk), _ptr(y), _ptr(h), _ptr(state), _ptr(xn), _ptr(stats), _ptr(ws), nbytes,
                             _ptr(tk), B, C, N, Hs, stage, G, st), "kmu_mixer_fwd_stage")
        ctx.save_for_backward(x, lw, stats, xn, w_bcdt, w_dw, w_hz, w_out, D, state)
        ctx.set_materialize_grads(False)
        ctx.dims = (B, C, N, Hs)
        ctx.zero_A = ops._const_zeros(A)
        ctx.lnshape = ln_w.shape
        return y, xin.view_as(xin)

    @staticmethod
    def backward(ctx, dy, dalias):
        lib = _lib.load()
        x, lw, stats, xn, w_bcdt, w_dw, w_hz, w_out, D, state = ctx.saved_tensors
        B, C, N, Hs = ctx.dims
        L = Hs * Hs
        dev = x.device
        if dy is None:
            return (dalias,) + (None,) * 9
        dxn, parts = ops._mixer_backward(xn, dy, None, w_bcdt, w_dw, w_hz, w_out, D, state, ctx.dims, ctx.pack_ok, groups=G, tag="_g")
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        d_bcdt, d_dw, d_hz, d_out, d_D = mk(G, 3 * N, C), mk(G, 3 * N, 9), mk(G, 2 * C * C), mk(G, C * C), mk(G, 1)
        Bs = B // G     # partial rows are (sample b, tile / gate block / channel slice): the group of a row is b % G
        pairs = []
        for g in range(G):
            pairs += ops._mixer_colsum_pairs(parts, (d_bcdt[g], d_dw[g], d_hz[g], d_out[g], d_D[g]), N,
                                             grp=lambda p, g=g: p.view(Bs, G, p.shape[0] // B, *p.shape[1:])[:, g])
        _wgrad(lambda: ops.colsum(*[p for p, _ in pairs], outs=[o for _, o in pairs]), ctx.defer_wgrad)
        st = _stream()
        # LayerNorm1D backward, the blend partner's gradient added in its epilogue
        addend = None if dalias is None else _f32c(dalias, "grad of the alias")
        rows = lib.kmu_layernorm1d_partials(B, C, L)
        dx = torch.empty_like(x)
        dwp = torch.empty(rows, C, device=dev, dtype=torch.float32)
        dbp = torch.empty(rows, C, device=dev, dtype=torch.float32)
        _lib.check(_call(("layernorm1d_bwd_g", (B, C, L)), lib.kmu_layernorm1d_bwd_add, _ptr(x), _ptr(lw), _ptr(stats), _ptr(dxn), _ptr(addend),
                         _ptr(dx), _ptr(dwp), _ptr(dbp), B, C, L, G, st), "kmu_layernorm1d_bwd_add")
        dw, db = mk(G, C), mk(G, C)
        nb = rows // B                      # partial rows are (sample b, block): group of a row = b % G

        lgrp = lambda p, g: p.view(B // G, G, nb, C)[:, g]
        _wgrad(lambda: ops.colsum(*[lgrp(p, g) for g in range(G) for p in (dwp, dbp)], outs=[o[g] for g in range(G) for o in (dw, db)]),
               ctx.defer_ln)
        return (dx, dw.view(ctx.lnshape), db.view(ctx.lnshape), None, d_bcdt.view(G * 3 * N, C, 1), d_dw.view(G * 3 * N, 1, 3, 3),
                d_hz.view(G * 2 * C, C, 1), d_out.view(G * C, C, 1), ctx.zero_A, d_D.view(G))


# ------------------------------------------------------------------------------------------ pointwise convs, grouped
def _pw_fwd_g(lib, x, w, bias, ci, co, act_in=0):
    B, _, H, W = x.shape
    y = torch.empty(B, G * co, H, W, device=x.device, dtype=torch.float32)
    _lib.check(_call(("pwconv_fwd_g", (B, ci, co, H * W)), lib.kmu_pwconv_fwd_g, _ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, ci, co, H * W,
                     int(act_in), G, _stream()), "kmu_pwconv_fwd_g")
    return y


def _pw_dgrad_g(lib, gy, w, ci, co, addend=None):
    B, _, H, W = gy.shape
    dx = torch.empty(B, G * ci, H, W, device=gy.device, dtype=torch.float32)
    _lib.check(_call(("pwconv_bwd_input_g", (B, ci, co, H * W)), lib.kmu_pwconv_bwd_input_g, _ptr(gy), _ptr(w), None, _ptr(addend), _ptr(dx),
                     B, ci, co, H * W, 0, G, _stream()), "kmu_pwconv_bwd_input_g")
    return dx


def _pw_wgrad_g(lib, x, gy, dw, db, ci, co):
    """dw [G*co, ci], db [G*co] or None: one (wgrad + slab reduce) pair per group; runs inside a deferred job"""
    B, _, H, W = x.shape
    P = H * W
    nbytes = lib.kmu_pwconv_bwd_weight_ws_bytes(B, ci, co, P)
    for g in range(G):
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        _lib.check(_call(("pwconv_bwd_weight_g", (B, ci, co, P)), lib.kmu_pwconv_bwd_weight_g, _ptr(x), _ptr(gy), _ptr(dw[g * co:(g + 1) * co]),
                         _ptr(None if db is None else db[g * co:(g + 1) * co]), _ptr(ws), nbytes, B, ci, co, P, 0, G, g, _stream()),
                   "kmu_pwconv_bwd_weight_g")


class PwConvGFn(torch.autograd.Function):
    """Block-diagonal 1x1 conv with bias: x [B, G*Ci, H, W] -> [B, G*Co, H, W], weight [G*Co, Ci, 1, 1] (DirectionAttention.qkv of
    the three branches, KM_UNetV3_SH.py:221)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        ctx.defer_wgrad = _leaf(weight, bias)
        x = _f32c(x, "x")
        co, ci = weight.shape[0] // G, weight.shape[1]
        w = _f32c(weight, "weight").view(G * co, ci)
        y = _pw_fwd_g(lib, x, w, None if bias is None else _f32c(bias, "bias"), ci, co)
        ctx.save_for_backward(x, w)
        ctx.cfg = (bias is not None, ci, co, tuple(weight.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        has_bias, ci, co, wshape = ctx.cfg
        g = _f32c(g, "grad")
        dx = _pw_dgrad_g(lib, g, w, ci, co) if ctx.needs_input_grad[0] else None
        dw = torch.empty(G * co, ci, device=x.device, dtype=torch.float32)
        db = torch.empty(G * co, device=x.device, dtype=torch.float32) if has_bias else None
        _wgrad(lambda: _pw_wgrad_g(lib, x, g, dw, db, ci, co), ctx.defer_wgrad)
        return dx, dw.view(wshape), db


class FfnBlendGFn(torch.autograd.Function):
    """ops.FfnBlendFn for the stacked branches: x + sigmoid(a) (BN2(fc2(ReLU(BN1(fc1(x))))) - x) with block-diagonal fc1 / fc2
    (efficient_vim_init.py:96; vim_utils_init.py:62-89,122-130); BatchNorm is per channel and runs unchanged on G*C channels."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, mom1, eps1, w2, g2, b2, rm2, rv2, mom2, eps2, a_row, training):
        lib = _lib.load()
        ctx.defer_wgrad = _leaf(w1, w2)
        x, a_row = _f32c(x, "x"), _f32c(a_row, "alpha row")
        hid, C = w1.shape[0] // G, w1.shape[1]
        w1c, w2c = _f32c(w1, "fc1 weight").view(G * hid, C), _f32c(w2, "fc2 weight").view(G * C, hid)
        z1 = _pw_fwd_g(lib, x, w1c, None, C, hid)
        h, st1 = ops._k_bn_fwd(lib, z1, None, g1, b1, None, rm1, rv1, mom1, eps1, 1, training, None, tap_groups=G)
        z2 = _pw_fwd_g(lib, h, w2c, None, hid, C)
        out, st2 = ops._k_bn_fwd(lib, z2, x, g2, b2, a_row, rm2, rv2, mom2, eps2, 0, training, None)
        ctx.save_for_backward(x, w1c, z1, st1, h, w2c, z2, st2, g1, b1, g2, b2, a_row)
        ctx.cfg = (int(training), tuple(w1.shape), tuple(w2.shape), hid, C)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w1, z1, st1, h, w2, z2, st2, g1, b1, g2, b2, a_row = ctx.saved_tensors
        training, s1, s2, hid, C = ctx.cfg
        dz2, dxb, dg2, db2, da = ops._k_bn_bwd(lib, _f32c(g, "grad"), z2, x, g2, b2, a_row, st2, 0, training)
        dh = _pw_dgrad_g(lib, dz2, w2, hid, C)
        dw2 = torch.empty(G * C, hid, device=x.device, dtype=torch.float32)
        _wgrad(lambda: _pw_wgrad_g(lib, h, dz2, dw2, None, hid, C), ctx.defer_wgrad)
        dz1, _, dg1, db1, _ = ops._k_bn_bwd(lib, dh, z1, None, g1, b1, None, st1, 1, training)
        dx = _pw_dgrad_g(lib, dz1, w1, C, hid, addend=dxb)
        dw1 = torch.empty(G * hid, C, device=x.device, dtype=torch.float32)
        _wgrad(lambda: _pw_wgrad_g(lib, x, dz1, dw1, None, C, hid), ctx.defer_wgrad)
        return (dx, dw1.view(s1), dg1, db1, None, None, None, None, dw2.view(s2), dg2, db2, None, None, None, None, da, None)


# ------------------------------------------------------------------------------------------ gate MLP, grouped
class GateMlpGFn(torch.autograd.Function):
    """G gate MLPs on p [B, G*I] -> g [B, G*O] with stacked weights (DirectionAttention.fc, KM_UNetV3_SH.py:231-236)."""

    @staticmethod
    def forward(ctx, p, w1, b1, w2, b2, act1, act2):
        lib = _lib.load()
        p = _f32c(p, "pooled input")
        B, I = p.shape[0], p.shape[1] // G
        H, O = w1.shape[0] // G, w2.shape[0] // G
        w1c, w2c = _f32c(w1, "w1").view(G * H, I), _f32c(w2, "w2").view(G * O, H)
        z1 = torch.empty(B, G * H, device=p.device, dtype=torch.float32)
        g = torch.empty(B, G * O, device=p.device, dtype=torch.float32)
        _lib.check(_call(("gate_mlp_fwd_g", (B, I, H, O)), lib.kmu_gate_mlp_fwd_g, _ptr(p), _ptr(w1c), _ptr(_f32c(b1, "b1")), _ptr(w2c),
                         _ptr(_f32c(b2, "b2")), _ptr(z1), _ptr(g), B, I, H, O, ops._ACT1[act1], ops._ACT2[act2], G, _stream()),
                   "kmu_gate_mlp_fwd_g")
        ctx.save_for_backward(p, w1c, w2c, z1, g)
        ctx.cfg = (ops._ACT1[act1], ops._ACT2[act2], tuple(w1.shape), tuple(w2.shape), I, H, O)
        return g

    @staticmethod
    def backward(ctx, dg):
        lib = _lib.load()
        p, w1, w2, z1, g = ctx.saved_tensors
        a1, a2, s1, s2, I, H, O = ctx.cfg
        dg = _f32c(dg, "grad")
        B, dev = p.shape[0], p.device
        dp = torch.empty_like(p)
        dw1, dw2 = torch.empty(G * H, I, device=dev), torch.empty(G * O, H, device=dev)
        db1, db2 = torch.empty(G * H, device=dev), torch.empty(G * O, device=dev)
        _lib.check(_call(("gate_mlp_bwd_g", (B, I, H, O)), lib.kmu_gate_mlp_bwd_g, _ptr(p), _ptr(w1), _ptr(w2), _ptr(z1), _ptr(g), _ptr(dg),
                         _ptr(dp), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), B, I, H, O, a1, a2, G, _stream()), "kmu_gate_mlp_bwd_g")
        return dp, dw1.view(s1), db1, dw2.view(s2), db2, None, None


# ------------------------------------------------------------------------------------------ fusion gate + mix on the stacked tensor
class GatedMix3StackedFn(torch.autograd.Function):
    """ops.GatedMix3Fn with the three branch outputs as channel slices of F [B, 3C, H, W] (KM_UNetV3_SH.py:111-117, :141-146)."""

    @staticmethod
    def forward(ctx, x, F, w1, b1, w2, b2, s):
        lib = _lib.load()
        x, F = _f32c(x, "x"), _f32c(F, "F")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        n = C * HW
        dev, st = x.device, _stream()
        Hd = w1.shape[0]
        w1c, w2c = _f32c(w1, "w1").view(Hd, 3 * C), _f32c(w2, "w2").view(3, Hd)
        b1c, b2c = _f32c(b1, "b1"), _f32c(b2, "b2")
        pooled = torch.empty(B, 3 * C, device=dev, dtype=torch.float32)
        _lib.check(_call(("mean_rows", (B, 3 * C, HW)), lib.kmu_mean_rows, _ptr(F), None, None, _ptr(pooled), B, 3 * C, HW, 1, st),
                   "kmu_mean_rows")
        z1 = torch.empty(B, Hd, device=dev, dtype=torch.float32)
        g = torch.empty(B, 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("gate_mlp_fwd", (B, 3 * C, Hd, 3)), lib.kmu_gate_mlp_fwd, _ptr(pooled), _ptr(w1c), _ptr(b1c), _ptr(w2c), _ptr(b2c),
                         _ptr(z1), _ptr(g), B, 3 * C, Hd, 3, ops._ACT1["gelu"], ops._ACT2["softmax"], st), "kmu_gate_mlp_fwd")
        sc = None if s is None else _f32c(s, "s").view(B)
        out = torch.empty_like(x)
        _lib.check(_call(("mix3_fwd_stacked", (B, n)), lib.kmu_mix3_fwd_stacked, _ptr(x), _ptr(F), _ptr(g), _ptr(sc), _ptr(out), B, n, st),
                   "kmu_mix3_fwd_stacked")
        ctx.save_for_backward(F, g, sc, pooled, w1c, w2c, z1)
        ctx.cfg = (B, C, HW, Hd, tuple(w1.shape), tuple(w2.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        F, g, sc, pooled, w1c, w2c, z1 = ctx.saved_tensors
        B, C, HW, Hd, s1, s2 = ctx.cfg
        n = C * HW
        dy = _f32c(dy, "dy")
        dev, st = dy.device, _stream()
        part = torch.empty(lib.kmu_mix3_blocks(n), B * 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("mix3_bwd_dg_stacked", (B, n)), lib.kmu_mix3_bwd_dg_stacked, _ptr(dy), _ptr(F), _ptr(g), _ptr(sc), _ptr(part), B, n,
                         st), "kmu_mix3_bwd_dg_stacked")
        (dg,) = colsum(part)
        dpool = torch.empty_like(pooled)
        dw1, dw2 = torch.empty(Hd, 3 * C, device=dev), torch.empty(3, Hd, device=dev)
        db1, db2 = torch.empty(Hd, device=dev), torch.empty(3, device=dev)
        _lib.check(_call(("gate_mlp_bwd", (B, 3 * C, Hd, 3)), lib.kmu_gate_mlp_bwd, _ptr(pooled), _ptr(w1c), _ptr(w2c), _ptr(z1), _ptr(g),
                         _ptr(dg), _ptr(dpool), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), B, 3 * C, Hd, 3, ops._ACT1["gelu"],
                         ops._ACT2["softmax"], st), "kmu_gate_mlp_bwd")
        dF = torch.empty_like(F)
        _lib.check(_call(("mix3_bwd_apply_stacked", (B, n)), lib.kmu_mix3_bwd_apply_stacked, _ptr(dy), _ptr(g), _ptr(sc), _ptr(dpool), _ptr(dF),
                         B, C, HW, st), "kmu_mix3_bwd_apply_stacked")
        return dy, dF, dw1.view(s1), db1, dw2.view(s2), db2, None


# ------------------------------------------------------------------------------------------ the stacked branch pass
def _bn_stacked(bns):
    """(gamma, beta, running_mean, running_var, momentum, eps, training) of G BatchNorm2d modules as [G*C] tensors."""
    b0 = bns[0]
    return (stack_params([b.weight for b in bns]), stack_params([b.bias for b in bns]), stacked_buffer([b.running_mean for b in bns]),
            stacked_buffer([b.running_var for b in bns]), b0.momentum, b0.eps, b0.training)


def supported(blocks, x):
    """The stacked pass covers the configuration KM-UNet builds: bf16x3 K2 kernels, channel counts the pointwise-conv kernels take,
    identical hyper-parameters in the three branches."""
    C, HW = x.shape[1], x.shape[2] * x.shape[3]
    e = blocks[0].vit_mamba
    hid = e.ffn.fc1.conv.out_channels
    return (x.is_cuda and ops.K2_MATH != "f32" and C in (16, 32, 64) and HW % 64 == 0 and int(round(HW ** 0.5)) ** 2 == HW and
            ops.pwconv_supported(C, hid, HW) and ops.pwconv_supported(C, 3 * C, HW) and
            all(b.vit_mamba.ffn.fc1.conv.out_channels == hid and b.vit_mamba.mixer.state_dim == 64 for b in blocks))


def direction_branches(blocks, xs):
    """blocks: the three DirectionViM modules; xs: their projected inputs [B, C, H, W].  Returns F [B, 3C, H, W] =
    cat_t attn_t(vit_mamba_t(xs_t)) computed as one stacked pass."""
    lib = _lib.load()
    ev = [b.vit_mamba for b in blocks]
    at = [b.attn for b in blocks]
    B, C, H, W = xs[0].shape
    x = torch.cat(xs, dim=1)                                             # [B, 3C, H, W]
    # ---- EfficientViMBlock x 3 (efficient_vim_init.py:81-97)
    A4 = stack_params([e.alpha for e in ev]).view(G, 4, C).permute(1, 0, 2).contiguous().view(4, G * C)      # rows a0..a3, [3C] each
    training = ev[0].training
    if training:
        nbt = [bn.num_batches_tracked for e in ev for bn in (e.dwconv1.norm, e.dwconv2.norm, e.ffn.fc1.norm, e.ffn.fc2.norm)
               if bn.track_running_stats and bn.num_batches_tracked is not None]
        if nbt:
            torch._foreach_add_(nbt, 1)
    g, bta, rm, rv, mom, eps, tr = _bn_stacked([e.dwconv1.norm for e in ev])
    x = ops.DwBnBlendFn.apply(x, stack_params([e.dwconv1.conv.weight for e in ev]), g, bta, A4[0], rm, rv, mom, eps, tr, None)
    ln_w, ln_b = stack_params([e.norm.weight for e in ev]), stack_params([e.norm.bias for e in ev])
    mx = [e.mixer for e in ev]
    mw = (stack_params([m.BCdt_proj.conv.weight for m in mx]), stack_params([m.dw.conv.weight for m in mx]),
          stack_params([m.hz_proj.conv.weight for m in mx]), stack_params([m.out_proj.conv.weight for m in mx]),
          stack_params([m.A for m in mx]), stack_params([m.D for m in mx]))
    if ops.K2_MATH == "v2":          # LayerNorm1D + HSMSSD as two launches (csrc/hsmssd_v2.inc); the blend takes the alias of x
        y, xa = MixerGFn.apply(x.view(B * G, C, H * W), ln_w, ln_b, ev[0].norm.eps, *mw)
        x = ops.bn_blend(y.view(B, G * C, H, W), xa.view_as(x), None, A4[1])
    else:
        y = HsmssdGFn.apply(LayerNorm1dGFn.apply(x.view(B * G, C, H * W), ln_w, ln_b, ev[0].norm.eps), *mw)
        x = ops.bn_blend(y.view(B, G * C, H, W), x, None, A4[1])
    g, bta, rm, rv, mom, eps, tr = _bn_stacked([e.dwconv2.norm for e in ev])
    x = ops.DwBnBlendFn.apply(x, stack_params([e.dwconv2.conv.weight for e in ev]), g, bta, A4[2], rm, rv, mom, eps, tr, None)
    g1, b1, rm1, rv1, mom1, eps1, tr = _bn_stacked([e.ffn.fc1.norm for e in ev])
    g2, b2, rm2, rv2, mom2, eps2, _ = _bn_stacked([e.ffn.fc2.norm for e in ev])
    x = FfnBlendGFn.apply(x, stack_params([e.ffn.fc1.conv.weight for e in ev]), g1, b1, rm1, rv1, mom1, eps1,
                          stack_params([e.ffn.fc2.conv.weight for e in ev]), g2, b2, rm2, rv2, mom2, eps2, A4[3], tr)
    # ---- DirectionAttention x 3 (KM_UNetV3_SH.py:215-263)
    gate = GateMlpGFn.apply(ops.spatial_mean(x), stack_params([a.fc[0].weight for a in at]), stack_params([a.fc[0].bias for a in at]),
                            stack_params([a.fc[2].weight for a in at]), stack_params([a.fc[2].bias for a in at]), "gelu", "sigmoid")
    qkv = PwConvGFn.apply(x, stack_params([a.qkv.weight for a in at]), stack_params([a.qkv.bias for a in at]))
    attn = ops.qkv_gate(qkv.view(B * G, 3 * C, H, W)).view(B, G * C, H, W)
    return ops.dwconv3x3_scaled(attn, stack_params([a.conv.weight for a in at]), stack_params([a.conv.bias for a in at]), gate)
