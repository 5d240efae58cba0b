"""torch.autograd.Function wrappers that launch the HIP kernels through the C ABI.

Every function takes/returns CUDA(ROCm) fp32 tensors and launches on torch's current stream,
so calls are stream-ordered with the surrounding PyTorch glue and hipGraph-capturable.  A CPU
tensor (or a missing library) raises: the hot path has no fallback.
"""
import os

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
_PROF = None


def profile_begin():
    global _PROF
    _PROF = {}


def profile_end():
    """-> {(entry point, shape): [ms per launch]} ; synchronises."""
    global _PROF
    prof, _PROF = _PROF, None
    torch.cuda.synchronize()
    return {k: [s.elapsed_time(e) for s, e in v] for k, v in (prof or {}).items()}


def _call(key, fn, *args):
    if _PROF is None:
        return fn(*args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn(*args)
    e.record()
    _PROF.setdefault(key, []).append((s, e))
    return rc


# Test support (tests/, smoke()): when set to a list, every op that applies a ReLU appends its branch mask (output > 0), in
# execution order.  The parity checks hand these masks to the oracle so that both sides differentiate the SAME piecewise-linear
# branch (oracle/ties.py: a pre-activation within rounding of zero may take either branch in two correct implementations).
RELU_TAP = None


def _tap_relu(t, groups=1):
    """groups > 1: t stacks the outputs of `groups` reference modules along the channel axis (grouped.py); the reference runs them
    one after the other, so their masks are reported separately, in that order."""
    if RELU_TAP is not None:
        for part in (t.detach() > 0).cpu().chunk(groups, dim=1):
            RELU_TAP.append(part)


def _f32c(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s: %s is on %s; the KM-UNet hot path only runs as HIP kernels on an MI355X "
                           "(no CPU fallback)" % ("kmunet", name, t.device))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---- weight gradients off the critical path ---------------------------------------------------------------------------------
# The backward pass is one dependent chain of activation-gradient kernels; every parameter-gradient kernel (1x1 / 3x3 / depthwise
# weight gradients, their slab reductions, column sums) hangs off that chain as a leaf that nothing waits for until the optimizer
# runs -- ~2.8 ms of ~17 ms kernel time per step at B = 8, in launches far too small to fill 256 CUs.  With WGRAD_OVERLAP set
# (DataParallel.backward does) such a leaf is not launched where autograd reaches it: `_wgrad(job)` queues the closure (its
# outputs are allocated at once and handed to autograd unfilled -- nothing reads a parameter gradient before the bucket copy),
# and every WGRAD_BATCH jobs the queue is dealt round-robin onto a few side streams that first wait for the streams the jobs'
# inputs were produced on.  flush_wgrad_jobs(final=True) drains the queue and makes the current stream wait for the side streams.
# Forking per job instead (one event per weight gradient, ~300 per step) was measured SLOWER than the serial order inside the
# captured graph (16.9 vs 15.05 ms/step): every cross-stream edge costs the main chain more than the ~5 us leaf it moves.
# Off by default: a bare `.backward()` followed by a read of `.grad` must see finished gradients.
WGRAD_OVERLAP = False
WGRAD_BATCH = 4096      # jobs per mid-backward flush: larger than a step's queue -- one flush at the end measured best (32..512: +0.5 ms)
WGRAD_STREAMS = 4       # side lanes of the flush (3 / 4 / 5 within +-0.07 ms)
_WG_SIDE = {}
_WG_JOBS = []
_WG_BUSY = set()
# While the final flush runs its jobs, the second-stage reductions they would launch (column sums of partial rows, slab sums of
# the pointwise-conv weight gradients) are only REGISTERED here and then issued as a handful of multi-problem launches on the
# joining stream: the tail is bound by its launch count (~500 launches on three streams = 1.45 ms of a 12.1 ms step; the
# same step with the jobs skipped takes 10.6 ms), and these two kinds are a third of it.
_WG_BATCH = None


def _leaf(*ts):
    """True when every given tensor is a leaf (a parameter handed to the op as it is).  Only then may its gradient be filled in
    later: a weight that reaches the op through autograd operations (conv3tap's permuted taps, TripleNorm's summed affine
    parameters) has its gradient READ by those operations' backward right after the node returns.  grouped.StackParamsFn is the
    exception: its backward only slices the stacked gradient into views."""
    return all(t is None or t.grad_fn is None or type(t.grad_fn).__name__ == "StackParamsFnBackward" for t in ts)


def _wgrad(job, defer=True, heavy=False):
    if not (WGRAD_OVERLAP and defer):
        job()
        return
    _WG_JOBS.append((torch.cuda.current_stream(), job))
    if len(_WG_JOBS) >= WGRAD_BATCH:
        flush_wgrad_jobs(final=False)


def flush_wgrad_jobs(final=True):
    """Launch the queued weight-gradient jobs on the side streams; final: also make the current stream wait for them."""
    cur = torch.cuda.current_stream()
    batch = None
    if _WG_JOBS:
        dev = cur.device
        sides = _WG_SIDE.get(dev.index)
        if sides is None or len(sides) != WGRAD_STREAMS:
            sides = _WG_SIDE[dev.index] = [torch.cuda.Stream(device=dev) for _ in range(WGRAD_STREAMS)]
        producers = {}
        for st, _ in _WG_JOBS:
            producers[st.cuda_stream] = st
        events = [st.record_event() for st in producers.values()]
        for side in sides:
            for ev in events:
                side.wait_event(ev)
            _WG_BUSY.add(side)
        global _WG_BATCH
        batch = _WG_BATCH = {"colsum": [], "pwred": [], "pw": [], "bsum": []} if final else None
        try:
            for i, (_, job) in enumerate(_WG_JOBS):
                with torch.cuda.stream(sides[i % len(sides)]):
                    job()
            if batch and batch["pw"]:
                _issue_pw_partials(batch, sides)
        finally:
            _WG_BATCH = None
        _WG_JOBS.clear()        # inputs die here: their blocks go back to the producers' pools, whose next kernels are ordered
                                # behind the join below (main) or behind the next forward's fork from main (branch streams)
    if final:
        for side in _WG_BUSY:
            cur.wait_stream(side)
        _WG_BUSY.clear()
        if batch:
            _issue_batched(batch)       # on the joining stream, behind every job's first stage


def _cs_desc(p, o):
    """(rows, cols, row stride, rows per run, run stride) of partial array p whose column sum is o: p = [rows, *o.shape] packed, a column
    range p_full[:, a:b] of a packed array, or [runs, rows per run, *o.shape] (one weight group of a grouped launch's partials)."""
    cols = max(1, o.numel())
    if p.shape[0] * cols == p.numel():
        if p.shape[0] > 1 and not p[0].is_contiguous():
            raise RuntimeError("colsum: the rows of a partial array must be contiguous")
        return p.shape[0], cols, (int(p.stride(0)) if p.shape[0] > 1 else cols), p.shape[0], 0
    if p.dim() < 3 or p.shape[0] * p.shape[1] * cols != p.numel() or not p[0, 0].is_contiguous():
        raise RuntimeError("colsum: partial %s (strides %s) against result %s" % (tuple(p.shape), tuple(p.stride()), tuple(o.shape)))
    return p.shape[0] * p.shape[1], cols, (int(p.stride(1)) if p.shape[1] > 1 else cols), p.shape[1], int(p.stride(0))


def _cs_args(pairs):
    """ctypes argument block of kmu_colsum_multi_strided for (partial, result) pairs (built before the launch is timed)."""
    import ctypes
    n = len(pairs)
    vp, ip, lp = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_longlong * n
    d = [_cs_desc(p, o) for p, o in pairs]
    return (n, vp(*[p.data_ptr() for p, _ in pairs]), vp(*[o.data_ptr() for _, o in pairs]), ip(*[x[0] for x in d]), ip(*[x[1] for x in d]),
            ip(*[x[2] for x in d]), ip(*[x[3] for x in d]), lp(*[x[4] for x in d]))


def colsum(*partials, outs=None):
    """Column sums of per-workgroup partial arrays [rows, ...] -> [...] for up to 8 arrays in ONE launch
    (csrc/colsum.hip): the second stage of every deterministic two-stage parameter-gradient reduction.
    outs: preallocated results, one per non-None partial (deferred weight-gradient jobs)."""
    import ctypes
    lib = _lib.load()
    parts = [p for p in partials if p is not None]
    if outs is None:
        outs = [torch.empty(p.shape[1:], device=p.device, dtype=torch.float32) for p in parts]
    if _WG_BATCH is not None:
        _WG_BATCH["colsum"].extend(zip(parts, outs))
        it = iter(outs)
        return [None if p is None else next(it) for p in partials]
    args = _cs_args(list(zip(parts, outs)))
    _lib.check(_call(("colsum_multi", tuple(int(p.numel()) for p in parts)), lib.kmu_colsum_multi_strided, *args, _stream()),
               "kmu_colsum_multi")
    it = iter(outs)
    return [None if p is None else next(it) for p in partials]


def copy_multi(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for contiguous fp32 CUDA tensors of equal numel, 160 per launch (csrc/colsum.hip: kmu_copy_multi)."""
    import ctypes
    n = len(dsts)
    if n == 0:
        return
    for d, s_ in zip(dsts, srcs):
        if not (d.is_cuda and s_.is_cuda and d.dtype == s_.dtype == torch.float32 and d.is_contiguous() and s_.is_contiguous()
                and d.numel() == s_.numel()):
            raise RuntimeError("copy_multi: contiguous fp32 CUDA tensors of equal size only")
    keep = [(d, s_) for d, s_ in zip(dsts, srcs) if d.numel() > 0 and d.data_ptr() != s_.data_ptr()]
    if not keep:
        return
    n = len(keep)
    sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in keep])
    dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in keep])
    ne = (ctypes.c_longlong * n)(*[d.numel() for d, _ in keep])
    _lib.check(_call(("copy_multi", (n,)), _lib.load().kmu_copy_multi, n, sp, dp, ne, _stream()), "kmu_copy_multi")


class FanoutFn(torch.autograd.Function):
    """n aliases of x for n consumers; the backward sums their gradients in ONE launch (csrc/colsum.hip: kmu_add_n) where autograd's
    own accumulation runs n - 1 pairwise adds.  Used where a tensor feeds parallel branches (KM_UNetV3_SH.py:141-146, :487-509)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if not gs[0].is_cuda or len(gs) > 4 or any(g.dtype != torch.float32 for g in gs):
            out = gs[0] + gs[1]
            for g in gs[2:]:
                out = out + g
            return out, None
        gs = [_f32c(g, "grad") for g in gs]
        out = torch.empty_like(gs[0])
        p = [_ptr(g) for g in gs] + [None] * (4 - len(gs))
        _lib.check(_call(("add_n", (len(gs), out.numel())), _lib.load().kmu_add_n, p[0], p[1], p[2], p[3], _ptr(out), out.numel(), _stream()),
                   "kmu_add_n")
        return out, None


def fanout(x, n):
    """n aliases of x whose gradients are summed by one kernel; plain aliases when no gradient is wanted."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return FanoutFn.apply(x, n)


def _pw_wgrad_call(lib, x, gy, dw, db, B, ci, co, P, act_in):
    """kmu_pwconv_bwd_weight, or -- inside the final flush -- only its slab pass, the slab reduction being batched"""
    nbytes = lib.kmu_pwconv_bwd_weight_ws_bytes(B, ci, co, P)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    if _WG_BATCH is None:
        _lib.check(_call(("pwconv_bwd_weight", (B, ci, co, P)), lib.kmu_pwconv_bwd_weight, _ptr(x), _ptr(gy), _ptr(dw), _ptr(db), _ptr(ws),
                         nbytes, B, ci, co, P, int(act_in), _stream()), "kmu_pwconv_bwd_weight")
        return
    # only registered: problems of identical dimensions (the same layer of the three branches / the two blocks of a level) share a
    # launch (_issue_pw_partials), and every slab reduction shares one (_issue_batched)
    _WG_BATCH["pw"].append((x, gy, ws, dw, db, B, ci, co, P, int(act_in)))


def _issue_pw_partials(batch, lanes):
    """slab passes of the registered pointwise-conv weight gradients, up to 8 identical problems per launch, dealt over `lanes`"""
    import ctypes
    lib = _lib.load()
    groups = {}
    for pr in batch["pw"]:
        groups.setdefault((pr[5], pr[6], pr[7], pr[8], pr[9], pr[4] is not None), []).append(pr)
    k = 0
    for (B, ci, co, P, act_in, has_b), prs in groups.items():
        nbytes = lib.kmu_pwconv_bwd_weight_ws_bytes(B, ci, co, P)
        for i in range(0, len(prs), 8):
            ch = prs[i:i + 8]
            n = len(ch)
            vp = ctypes.c_void_p * n
            with torch.cuda.stream(lanes[k % len(lanes)]):
                _lib.check(_call(("pwconv_bwd_weight", (B, ci, co, P)), lib.kmu_pwconv_bwd_weight_partial_multi, n,
                                 vp(*[c[0].data_ptr() for c in ch]), vp(*[c[1].data_ptr() for c in ch]), vp(*[c[2].data_ptr() for c in ch]),
                                 nbytes, int(has_b), B, ci, co, P, act_in, _stream()), "kmu_pwconv_bwd_weight_partial_multi")
            k += 1
            batch["pwred"].extend((c[2], c[3], c[4], B, ci, co, P) for c in ch)


def bias_sum(dy, db):
    """db[c] = sum of dy[b, c, ...] over batch and pixels (a convolution's bias gradient).  Inside the final weight-gradient flush the
    request is only registered: all of them go out as one kmu_bias_sum_multi launch on the joining stream."""
    if _WG_BATCH is not None and "bsum" in _WG_BATCH:
        _WG_BATCH["bsum"].append((dy, db))
        return
    _bias_sum_launch([(dy, db)])


def _bias_sum_launch(pairs):
    import ctypes
    lib, st = _lib.load(), _stream()
    for i in range(0, len(pairs), 32):
        ch = pairs[i:i + 32]
        n = len(ch)
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        _lib.check(_call(("bias_sum_multi", (n,)), lib.kmu_bias_sum_multi, n, vp(*[d.data_ptr() for d, _ in ch]), vp(*[o.data_ptr() for _, o in ch]),
                         ip(*[d.shape[0] for d, _ in ch]), ip(*[d.shape[1] for d, _ in ch]),
                         ip(*[d.numel() // (d.shape[0] * d.shape[1]) for d, _ in ch]), st), "kmu_bias_sum_multi")


def _issue_batched(batch):
    import ctypes
    lib = _lib.load()
    st = _stream()
    if batch.get("bsum"):
        _bias_sum_launch(batch["bsum"])
    red = batch["pwred"]
    for i in range(0, len(red), 32):
        ch = red[i:i + 32]
        n = len(ch)
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        _lib.check(lib.kmu_pwconv_bwd_weight_reduce_multi(n, vp(*[c[0].data_ptr() for c in ch]), vp(*[c[1].data_ptr() for c in ch]),
                                                          vp(*[None if c[2] is None else c[2].data_ptr() for c in ch]),
                                                          ip(*[c[3] for c in ch]), ip(*[c[4] for c in ch]), ip(*[c[5] for c in ch]),
                                                          ip(*[c[6] for c in ch]), st), "kmu_pwconv_bwd_weight_reduce_multi")
    cs = batch["colsum"]
    for i in range(0, len(cs), 64):
        ch = cs[i:i + 64]
        n = len(ch)
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        _lib.check(lib.kmu_colsum_multi_strided(*_cs_args(ch), st), "kmu_colsum_multi")


# ------------------------------------------------------------------------------------------ weight packs, once per step
# The split-bf16 kernels (K1, the plain KxK convolutions, K2) read fragment-ordered (hi, lo) packs of their weights.  Packing
# where the weight is used costs 63 tiny launches per training step (29 conv + 4 KAN input-gradient + 15 + 15 HSMSSD), each the
# head of a dependent chain.  Inside `pack_scope()` -- train.TrainStep opens one around forward + backward -- a pack made from a
# Parameter is kept in a persistent buffer and reused by every later consumer of the same step (K2's backward reuses the
# forward's); from the second step on `prepack()` refills ALL known packs with two launches (one job table per source file:
# kmu_conv_pack_multi, kmu_hsm_pack_multi) at the head of the step, so a captured hipGraph starts with them and every replay
# re-packs the current weights.  Outside a scope nothing is cached: a bare forward always packs the weights it is given.


class PackCache:
    def __init__(self):
        self.enabled = False
        self.epoch = 0
        self.entries = {}        # key -> [buffer, epoch packed, source tensors (kept alive: the key holds their addresses), job, epoch last asked for]
        self.plan = None
        self.dirty = False

    def clear(self):
        self.entries.clear()
        self.plan, self.dirty = None, False

    def build_plan(self, device):
        import numpy as np
        lib = _lib.load()
        rec = lib.kmu_pack_job_bytes()
        conv = [e for e in self.entries.values() if e[3][0] == "conv"]
        hsm = [e for e in self.entries.values() if e[3][0] == "hsm"]
        tabs = []
        for group in (conv, hsm):
            if not group:
                tabs.append(None)
                continue
            host = np.zeros(len(group) * rec, dtype=np.uint8)
            for i, (buf, _, srcs, job, _used) in enumerate(group):
                if job[0] == "conv":
                    _, which, Cin, Cout, K = job
                    ptrs = [_ptr(t) for t in srcs] + [None] * (3 - len(srcs))
                    _lib.check(lib.kmu_conv_pack_job(host.ctypes.data, i, which, ptrs[0], ptrs[1], ptrs[2], _ptr(buf), Cin, Cout, K),
                               "kmu_conv_pack_job")
                else:
                    _, C, groups = job
                    _lib.check(lib.kmu_hsm_pack_job(host.ctypes.data, i, _ptr(srcs[0]), _ptr(srcs[1]), _ptr(buf), C, groups), "kmu_hsm_pack_job")
            tabs.append((torch.from_numpy(host).to(device), len(group)))
        self.plan, self.dirty = tabs, False


_PACKS = PackCache()


class pack_scope:
    """with pack_scope(): one training step's forward + backward -- weights do not change inside."""

    def __enter__(self):
        self.outer = _PACKS.enabled
        if not self.outer:
            _PACKS.enabled = True
            _PACKS.epoch += 1
        return self

    def __exit__(self, *exc):
        _PACKS.enabled = self.outer
        return False


def prepack():
    """Refill every pack seen so far from the current weights (two launches); call at the head of a pack_scope."""
    pc = _PACKS
    if not pc.enabled or not pc.entries:
        return
    if not torch.cuda.is_current_stream_capturing():
        # packs nobody has asked for during the last 8 scopes belong to a model that is gone (or to weights that were replaced
        # out of place): drop them with their source tensors instead of re-packing them every step for ever
        dead = [k for k, e in pc.entries.items() if e[4] < pc.epoch - 8]
        for k in dead:
            del pc.entries[k]
        if dead:
            pc.dirty = True
            if not pc.entries:
                pc.plan = None
                return
    if pc.dirty or pc.plan is None:
        if torch.cuda.is_current_stream_capturing():      # the job table needs a host-to-device copy: packs stay lazy this step
            return
        pc.build_plan(next(iter(pc.entries.values()))[0].device)
    lib, st = _lib.load(), _stream()
    for tab, fn, nm in zip(pc.plan, (lib.kmu_conv_pack_multi, lib.kmu_hsm_pack_multi), ("conv_pack_multi", "hsm_pack_multi")):
        if tab is not None:
            _lib.check(_call((nm, (tab[1],)), fn, _ptr(tab[0]), tab[1], st), "kmu_" + nm)
    for e in pc.entries.values():
        e[1] = pc.epoch


def _pack_ok(origs, convs):
    """Cacheable: the op was handed Parameters (or grouped.StackParamsFn's zero-copy stack of adjacent Parameters) and reads them in
    place (fp32, contiguous): the weights keep their address and do not change inside a pack_scope."""
    def stable(o):
        return isinstance(o, torch.nn.Parameter) or (type(o.grad_fn).__name__ == "StackParamsFnBackward" and o._is_view())
    return all(stable(o) and o.data_ptr() == c.data_ptr() for o, c in zip(origs, convs))


def _packed(ok, srcs, nelems, job, pack_fn):
    """The bf16 pack buffer for `job` over the source tensors: cached per step inside a pack_scope, a fresh pack otherwise."""
    pc = _PACKS
    if not (ok and pc.enabled):
        buf = torch.empty(nelems, device=srcs[0].device, dtype=torch.bfloat16)
        pack_fn(buf)
        return buf
    key = job + tuple(t.data_ptr() for t in srcs)
    e = pc.entries.get(key)
    if e is None:
        if torch.cuda.is_current_stream_capturing():      # a persistent buffer must not come from the graph's private pool
            buf = torch.empty(nelems, device=srcs[0].device, dtype=torch.bfloat16)
            pack_fn(buf)
            return buf
        e = pc.entries[key] = [torch.empty(nelems, device=srcs[0].device, dtype=torch.bfloat16), -1, tuple(srcs), job, pc.epoch]
        pc.dirty = True
    e[4] = pc.epoch
    if e[1] != pc.epoch:
        pack_fn(e[0])
        e[1] = pc.epoch
    return e[0]


# ------------------------------------------------------------------------------------------ K1
import os as _os

# Matrix-core arithmetic of the 3x3 contractions (K1 and the plain 3x3 convolutions):
#   "bf16x3" (default): split-bf16 operands on v_mfma_f32_16x16x32_bf16 (csrc/conv3x3_x3.hip), ~1e-5 relative to the result
#   "f32"             : exact fp32 on v_mfma_f32_16x16x4_f32 (csrc/kan_conv2d.hip) -- the reference the bf16x3 kernels are tested against
K1_MATH = _os.environ.get("KMU_K1_MATH", "bf16x3")
# K2 forward: "bf16x3" = projection and depthwise 3x3 composed into one 3x3 convolution on the bf16 matrix core
# (csrc/hsmssd_x3.inc); "f32" = the exact-fp32 projection + LDS stencil kernels (csrc/hsmssd.hip)
#           "v2" (default) = round 4: LayerNorm + HSMSSD forward in two launches (csrc/hsmssd_v2.inc); its backward is the bf16x3 one
K2_MATH = _os.environ.get("KMU_K2_MATH", "v2")
_GRID_OK = {}


def _check_grid(grid):
    """KANLinear.grid is [in_features, 12] with identical rows (KANlayers.py:526-535).  The kernels
    share one knot vector; verified once per buffer version (host sync, so never inside a capture)."""
    key = (grid.data_ptr(), grid._version, tuple(grid.shape))
    if key not in _GRID_OK:
        if grid.dim() != 2 or grid.shape[1] != 12:
            raise RuntimeError("KANConv2d: expected a [in_features, 12] knot buffer, got %s" % (tuple(grid.shape),))
        if not bool((grid == grid[0:1]).all()):
            raise RuntimeError("KANConv2d: per-feature knot vectors differ (update_grid() was called?); "
                               "the HIP kernel supports one shared knot vector only")
        if not bool((grid[0, 1:] > grid[0, :-1]).all()):
            raise RuntimeError("KANConv2d: knots must be strictly increasing")
        # the bf16x3 kernel finds the knot span by one multiply + one correction step: needs knots within h/4 of uniform
        # (true for the layer's own grid, KANlayers.py:526-535; anything else runs on the exact-fp32 kernel's 12-compare search)
        u = grid[0].double().cpu()
        h = (u[11] - u[0]) / 11
        _GRID_OK[key] = bool(((u - (u[0] + h * torch.arange(12, dtype=torch.float64))).abs() < 0.25 * h).all())
    return grid[0].contiguous(), _GRID_OK[key]


class KanConv2dFn(torch.autograd.Function):
    """y = KANConv2d_3x3_s1_p1(x)  [+ residual] [ReLU]   (convKAN/KANConv2Dlayers.py:15-37)."""

    @staticmethod
    def forward(ctx, x, grid, base_w, spline_w, scaler, residual, relu):
        ctx.defer_wgrad = _leaf(base_w, spline_w, scaler)
        lib = _lib.load()
        x = _f32c(x, "x")
        origs = (base_w, spline_w, scaler)
        base_w, spline_w, scaler = _f32c(base_w, "base_weight"), _f32c(spline_w, "spline_weight"), _f32c(scaler, "spline_scaler")
        ctx.pack_ok = _pack_ok(origs, (base_w, spline_w, scaler))
        knots, uniform = _check_grid(grid)
        B, Cin, H, W = x.shape
        Cout = base_w.shape[0]
        if base_w.shape[1] != Cin * 9:
            raise RuntimeError("KANConv2d: base_weight %s does not match Cin=%d, 3x3" % (tuple(base_w.shape), Cin))
        need_bwd = any(ctx.needs_input_grad)
        x3 = K1_MATH == "bf16x3" and uniform and Cin % 4 == 0
        wp_f = None if x3 else torch.empty(lib.kmu_kan_pack_fwd_elems(Cin, Cout), device=x.device, dtype=torch.float32)
        wp_b = torch.empty(lib.kmu_kan_pack_bwd_elems(Cin, Cout), device=x.device, dtype=torch.float32) if (need_bwd and not x3) else None
        st = _stream()
        if wp_f is not None or wp_b is not None:
            _lib.check(lib.kmu_kan_pack_weights(_ptr(base_w), _ptr(spline_w), _ptr(scaler), _ptr(wp_f), _ptr(wp_b), Cin, Cout, st),
                       "kmu_kan_pack_weights")
        y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = _f32c(residual, "residual")
        if x3:
            wp3 = _packed(ctx.pack_ok, (base_w, spline_w, scaler), lib.kmu_conv3x3_x3_pack_elems(1, Cin, Cout), ("conv", 0, Cin, Cout, 3),
                          lambda buf: _lib.check(lib.kmu_kan_pack_weights_x3(_ptr(base_w), _ptr(spline_w), _ptr(scaler), _ptr(buf), Cin, Cout,
                                                                             st), "kmu_kan_pack_weights_x3"))
            _lib.check(_call(("kan_conv2d_fwd_x3", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_fwd_x3, _ptr(x), _ptr(knots), _ptr(wp3),
                             _ptr(residual), _ptr(y), B, Cin, Cout, H, W, 1 if relu else 0, st), "kmu_kan_conv2d_fwd_x3")
        else:
            _lib.check(_call(("kan_conv2d_fwd", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_fwd, _ptr(x), _ptr(knots), _ptr(wp_f),
                             _ptr(residual), _ptr(y), B, Cin, Cout, H, W, 1 if relu else 0, st), "kmu_kan_conv2d_fwd")
        if relu:
            _tap_relu(y)
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        ctx.x3 = x3
        ctx.save_for_backward(x, knots, spline_w, scaler, wp_b, y if relu else None, base_w if x3 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, knots, spline_w, scaler, wp_b, y, base_w = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        if ctx.relu:
            masked = torch.empty_like(dy)
            _lib.check(_call(("relu_mask", (dy.numel(),)), lib.kmu_relu_mask, _ptr(dy), _ptr(y), _ptr(masked), dy.numel(), _stream()),
                       "kmu_relu_mask")
            dy = masked
        B, Cin, H, W = x.shape
        Cout = dy.shape[1]
        st = _stream()
        dx = d_bw = d_sw = d_sc = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.x3:       # matrix core: conv3x3(dy, flipped W') per basis + the dPhi epilogue
                wpd = _packed(ctx.pack_ok, (base_w, spline_w, scaler), lib.kmu_kan_dgrad_x3_pack_elems(Cin, Cout), ("conv", 1, Cin, Cout, 3),
                              lambda buf: _lib.check(lib.kmu_kan_pack_weights_dgrad_x3(_ptr(base_w), _ptr(spline_w), _ptr(scaler), _ptr(buf),
                                                                                       Cin, Cout, st), "kmu_kan_pack_weights_dgrad_x3"))
                _lib.check(_call(("kan_conv2d_bwd_input_x3", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_bwd_input_x3, _ptr(x), _ptr(dy),
                                 _ptr(knots), _ptr(wpd), _ptr(dx), B, Cin, Cout, H, W, st), "kmu_kan_conv2d_bwd_input_x3")
            else:
                _lib.check(_call(("kan_conv2d_bwd_input", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_bwd_input, _ptr(x), _ptr(dy),
                                 _ptr(knots), _ptr(wp_b), _ptr(dx), B, Cin, Cout, H, W, st), "kmu_kan_conv2d_bwd_input")
        if any(ctx.needs_input_grad[2:5]):
            d_bw = torch.empty(Cout, Cin * 9, device=x.device, dtype=torch.float32)
            d_sw = torch.empty(Cout, Cin * 9, 8, device=x.device, dtype=torch.float32)
            d_sc = torch.empty(Cout, Cin * 9, device=x.device, dtype=torch.float32)
            x3 = ctx.x3

            def job():
                st = _stream()
                if x3:
                    nbytes = lib.kmu_conv3x3_x3_wgrad_ws_bytes(1, B, Cin, Cout, H, W)
                    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
                    _lib.check(_call(("kan_conv2d_bwd_weights_x3", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_bwd_weights_x3, _ptr(x), _ptr(dy),
                                     _ptr(knots), _ptr(spline_w), _ptr(scaler), _ptr(d_bw), _ptr(d_sw), _ptr(d_sc), _ptr(ws), nbytes,
                                     B, Cin, Cout, H, W, st), "kmu_kan_conv2d_bwd_weights_x3")
                else:
                    nbytes = lib.kmu_kan_bwd_ws_bytes(B, Cin, Cout, H, W)
                    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
                    _lib.check(_call(("kan_conv2d_bwd_weights", (B, Cin, Cout, H, W)), lib.kmu_kan_conv2d_bwd_weights, _ptr(x), _ptr(dy),
                                     _ptr(knots), _ptr(spline_w), _ptr(scaler), _ptr(d_bw), _ptr(d_sw), _ptr(d_sc), _ptr(ws), nbytes,
                                     B, Cin, Cout, H, W, st), "kmu_kan_conv2d_bwd_weights")
            _wgrad(job, ctx.defer_wgrad, heavy=True)
        return dx, None, d_bw, d_sw, d_sc, (dy if ctx.has_res else None), None


def kan_conv2d(x, grid, base_weight, spline_weight, spline_scaler, residual=None, relu=False):
    return KanConv2dFn.apply(x, grid, base_weight, spline_weight, spline_scaler, residual, relu)


class ConvKxKFn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, K, stride 1, padding K//2)(x), K in {3, 5, 7}, forward / input gradient / weight gradient on the
    split-bf16 matrix-core kernels (csrc/conv3x3_x3.hip): KM_UNetV3_SH.py:375 (conv_f), :430-446 (dec2[1], dec3[1], dec3[3]),
    :300-306 (MultiScaleFusion 3x3 / 5x5 / 7x7), DAGEM_md.py:43 (offset_conv)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.defer_wgrad = _leaf(weight, bias)
        lib = _lib.load()
        orig = weight
        x, weight = _f32c(x, "x"), _f32c(weight, "weight")
        ctx.pack_ok = _pack_ok((orig,), (weight,))
        bias = _f32c(bias, "bias") if bias is not None else None
        B, Cin, H, W = x.shape
        Cout, K = weight.shape[0], weight.shape[-1]
        if tuple(weight.shape) != (Cout, Cin, K, K) or K not in (3, 5, 7):
            raise RuntimeError("conv_kxk: weight %s does not match Cin=%d and a 3x3 / 5x5 / 7x7 kernel" % (tuple(weight.shape), Cin))
        st = _stream()
        wp = _packed(ctx.pack_ok, (weight,), lib.kmu_conv2d_x3_pack_elems(Cin, Cout, K), ("conv", 2, Cin, Cout, K),
                     lambda buf: _lib.check(lib.kmu_conv2d_pack_weights_x3(_ptr(weight), _ptr(buf), Cin, Cout, K, 0, st),
                                            "kmu_conv2d_pack_weights_x3"))
        y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
        _lib.check(_call(("conv%dx%d_fwd_x3" % (K, K), (B, Cin, Cout, H, W)), lib.kmu_conv2d_fwd_x3, _ptr(x), _ptr(wp), _ptr(bias), _ptr(y),
                         B, Cin, Cout, H, W, K, st), "kmu_conv2d_fwd_x3")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, Cin, H, W = x.shape
        Cout, K = weight.shape[0], weight.shape[-1]
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:       # dx = conv(dy, flipped / transposed weights) on the same kernel
            wp = _packed(ctx.pack_ok, (weight,), lib.kmu_conv2d_x3_pack_elems(Cout, Cin, K), ("conv", 3, Cin, Cout, K),
                         lambda buf: _lib.check(lib.kmu_conv2d_pack_weights_x3(_ptr(weight), _ptr(buf), Cin, Cout, K, 1, st),
                                                "kmu_conv2d_pack_weights_x3 (dgrad)"))
            dx = torch.empty_like(x)
            _lib.check(_call(("conv%dx%d_dgrad_x3" % (K, K), (B, Cin, Cout, H, W)), lib.kmu_conv2d_fwd_x3, _ptr(dy), _ptr(wp), None, _ptr(dx),
                             B, Cout, Cin, H, W, K, st), "kmu_conv2d_fwd_x3 (dgrad)")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cout, device=x.device, dtype=torch.float32)

        def job():
            if dw is not None:     # dW[o][c][tap] = sum_pix dy[o][pix] x[c][pix + tap - K//2]: transposed-read contraction
                nbytes = lib.kmu_conv2d_x3_wgrad_ws_bytes(B, Cin, Cout, H, W, K)
                ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
                _lib.check(_call(("conv%dx%d_bwd_weight_x3" % (K, K), (B, Cin, Cout, H, W)), lib.kmu_conv2d_bwd_weight_x3, _ptr(x), _ptr(dy),
                                 _ptr(dw), _ptr(ws), nbytes, B, Cin, Cout, H, W, K, _stream()), "kmu_conv2d_bwd_weight_x3")
            if db is not None:
                bias_sum(dy, db)
        if dw is not None or db is not None:
            _wgrad(job, ctx.defer_wgrad, heavy=True)
        return dx, dw, db


def conv_kxk(x, weight, bias=None):
    return ConvKxKFn.apply(x, weight, bias)


conv3x3 = conv_kxk


class ResizeBilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear", align_corners=True) (KM_UNetV3_SH.py:487-492, 503-507), csrc/resize.hip."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, Hi, Wi = x.shape
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        _lib.check(_call(("resize_bilinear_fwd", (B, C, Hi, Wi, Ho, Wo)), lib.kmu_resize_bilinear_ac_fwd, _ptr(x), _ptr(y), B, C, Hi, Wi,
                         Ho, Wo, _stream()), "kmu_resize_bilinear_ac_fwd")
        ctx.dims = (B, C, Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        B, C, Hi, Wi, Ho, Wo = ctx.dims
        gy = _f32c(gy, "gy")
        dx = torch.empty(B, C, Hi, Wi, device=gy.device, dtype=torch.float32)
        _lib.check(_call(("resize_bilinear_bwd", (B, C, Hi, Wi, Ho, Wo)), lib.kmu_resize_bilinear_ac_bwd, _ptr(gy), _ptr(dx), B, C, Hi, Wi,
                         Ho, Wo, _stream()), "kmu_resize_bilinear_ac_bwd")
        return dx, None, None


def resize_bilinear(x, size):
    return ResizeBilinearFn.apply(x, int(size[0]), int(size[1]))


# ------------------------------------------------------------------------------------------ K2
class LayerNorm1dFn(torch.autograd.Function):
    """Per-token LayerNorm over C of [B,C,L] (vim_utils_init.py:50-59)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, alias=False):
        lib = _lib.load()
        xin = x
        x = _f32c(x, "x")
        w, b = _f32c(weight, "weight").reshape(-1), _f32c(bias, "bias").reshape(-1)
        B, C, L = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B, L, 2, device=x.device, dtype=torch.float32)
        _lib.check(lib.kmu_layernorm1d_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(stats), B, C, L, float(eps), _stream()),
                   "kmu_layernorm1d_fwd")
        ctx.save_for_backward(x, w, stats)
        ctx.wshape = weight.shape
        ctx.defer_wgrad = _leaf(weight, bias)
        if alias:      # second output: x itself, for a second consumer whose gradient the backward kernel adds in (no fan-in launch)
            ctx.set_materialize_grads(False)
            return y, xin.view_as(xin)
        return y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        lib = _lib.load()
        x, w, stats = ctx.saved_tensors
        B, C, L = x.shape
        if dy is None:                   # only the alias was used
            return dalias, None, None, None, None
        dy = _f32c(dy, "dy")
        addend = None if dalias is None else _f32c(dalias, "grad of the alias")
        rows = lib.kmu_layernorm1d_partials(B, C, L)
        dx = torch.empty_like(x)
        dwp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        dbp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        _lib.check(lib.kmu_layernorm1d_bwd_add(_ptr(x), _ptr(w), _ptr(stats), _ptr(dy), _ptr(addend), _ptr(dx), _ptr(dwp), _ptr(dbp), B, C, L,
                                               1, _stream()), "kmu_layernorm1d_bwd_add")
        dw, db = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        _wgrad(lambda: colsum(dwp, dbp, outs=[dw, db]), ctx.defer_wgrad)
        return dx, dw.view(ctx.wshape), db.view(ctx.wshape), None, None


def layernorm1d(x, weight, bias, eps=1e-5):
    return LayerNorm1dFn.apply(x, weight, bias, eps)


def layernorm1d_alias(x, weight, bias, eps=1e-5):
    """(LayerNorm(x), x'): x' aliases x; its gradient is added inside the LayerNorm backward kernel."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return LayerNorm1dFn.apply(x, weight, bias, eps), x
    return LayerNorm1dFn.apply(x, weight, bias, eps, True)


def _hsm_pack(ok, w_bcdt, w_dw, C, st, groups=1):
    lib = _lib.load()
    return _packed(ok, (w_bcdt, w_dw), lib.kmu_hsmssd_pack_elems(C, groups), ("hsm", C, groups),
                   lambda buf: _lib.check(lib.kmu_hsmssd_pack_x3(_ptr(w_bcdt), _ptr(w_dw), _ptr(buf), C, groups, st), "kmu_hsmssd_pack_x3"))


def _hsmssd_backward(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dims, pack_ok, defer):
    """Backward of HSMSSD.forward given the input x, the saved gate state and dy / dh: three launches (pass A, gate,
    pass B) + the deferred column sums of the parameter-gradient partials.  -> dx, (d_bcdt, d_dw, d_hz, d_out, d_D)"""
    lib = _lib.load()
    B, C, N, Hs = dims
    dev = x.device
    dy = _f32c(dy, "dy") if dy is not None else torch.zeros(B, C, Hs, Hs, device=dev)
    dh = _f32c(dh, "dh") if dh is not None else None
    x3 = K2_MATH != "f32"
    P = (lib.kmu_hsmssd_bwd_partials_x3 if x3 else lib.kmu_hsmssd_bwd_partials)(B, C, Hs)
    dx = torch.empty_like(x)
    p_bcdt = torch.empty(P, 3 * N, C, device=dev, dtype=torch.float32)
    p_dw = torch.empty(P, 3 * N, 9, device=dev, dtype=torch.float32)
    G = lib.kmu_hsmssd_gate_partials(B)
    p_hz = torch.empty(G, 2 * C, C, device=dev, dtype=torch.float32)
    p_out = torch.empty(G, C, C, device=dev, dtype=torch.float32)
    p_D = torch.empty(G, device=dev, dtype=torch.float32)
    nbytes = (lib.kmu_hsmssd_bwd_ws_bytes_x3 if x3 else lib.kmu_hsmssd_bwd_ws_bytes)(B, C, N, Hs)
    ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
    st = _stream()
    fn, tail = lib.kmu_hsmssd_bwd_stage, (st,)
    if x3:
        fn, tail = lib.kmu_hsmssd_bwd_stage_x3_pk, (1, _ptr(_hsm_pack(pack_ok, w_bcdt, w_dw, C, st)), st)
    for stage, nm in enumerate(("hsmssd_bwd_passA", "hsmssd_bwd_gate", "hsmssd_bwd_passB")):   # one kernel per call
        _lib.check(_call((nm + ("_x3" if x3 and stage == 0 else ""), (B, C, Hs)), fn, _ptr(x), _ptr(dy), _ptr(dh), _ptr(w_bcdt), _ptr(w_dw),
                         _ptr(w_hz), _ptr(w_out), _ptr(D), _ptr(state), _ptr(dx), _ptr(p_bcdt), _ptr(p_dw), _ptr(p_hz),
                         _ptr(p_out), _ptr(p_D), _ptr(ws), nbytes, B, C, N, Hs, stage, *tail), "kmu_hsmssd_bwd_stage")
    mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    d_bcdt, d_dw, d_hz, d_out, d_D = mk(3 * N, C), mk(3 * N, 9), mk(2 * C, C), mk(C, C), mk(1)
    _wgrad(lambda: colsum(p_bcdt, p_dw, p_hz, p_out, p_D.view(G, 1), outs=[d_bcdt, d_dw, d_hz, d_out, d_D]), defer)
    return dx, (d_bcdt.view(3 * N, C, 1), d_dw.view(3 * N, 1, 3, 3), d_hz.view(2 * C, C, 1), d_out.view(C, C, 1), d_D.view(1))


_MIXER_BWD_STAGES = ("mixer_bwd_corr", "mixer_bwd_crows", "hsmssd_bwd_gate", "mixer_bwd_passB")
# Channel counts whose backward takes the C rows as a per-sample dense convolution (csrc/hsmssd_bwdc.inc).  Measured alone
# (tools/time_k2_bwd.py, profiles/r04_k2_backward_times.json), whole backward per mixer: C = 16 (8 x 128 x 128) 173 -> 146 us (four
# cheap chunks of pass B and the pass-A recompute go); C = 64 (24 x 32 x 32) 291 -> 254 us (pass B no longer keeps the dy image beside
# x: 108 -> 80 KB of LDS = two workgroups per CU, 221 -> 174 us; G = dy (*) x costs 9 C^2 floats per partial there, 28 MB of slabs);
# C = 32 (8 x 64 x 64) 105 -> 105 us with one launch more (256 tiles = one workgroup per CU either way) -- that level keeps pass A /
# pass B on all rows.  Every C is instantiated and tested on both routes.
MIXER_BWD_CROWS = (16, 64)


def _mixer_backward(xn, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, dims, pack_ok, groups=1, tag=""):
    """Backward behind MixerFn's forward given the (normalised) input, the saved gate state and dy / dh.  C in MIXER_BWD_CROWS:
    the C rows as a per-sample dense convolution with the M_b the forward left in `state` (correlation G = dy (*) x, the C rows'
    contractions of G, the gate stage, pass B on the {B, dt} rows); else pass A, gate, pass B on all rows.
    -> dx, (column-sum inputs, rows of w_bcdt / w_dw they belong to): see _mixer_colsums."""
    lib = _lib.load()
    B, C, N, Hs = dims
    dev = xn.device
    dy = _f32c(dy, "dy") if dy is not None else torch.zeros(B, C, Hs, Hs, device=dev)
    dh = _f32c(dh, "dh") if dh is not None else None
    P = lib.kmu_hsmssd_bwd_partials_x3(B, C, Hs)
    Gp = lib.kmu_hsmssd_gate_partials(B)
    mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    dx = torch.empty_like(xn)
    p_bcdt, p_dw = mk(P, 3 * N, C), mk(P, 3 * N, 9)
    p_hz, p_out, p_D = mk(Gp, 2 * C, C), mk(Gp, C, C), mk(Gp)
    st = _stream()
    if C not in MIXER_BWD_CROWS:
        nbytes = lib.kmu_hsmssd_bwd_ws_bytes_x3_g(B, C, N, Hs, groups)
        ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
        wpk = _hsm_pack(pack_ok, w_bcdt, w_dw, C, st, groups)
        for stage, nm in enumerate(("hsmssd_bwd_passA_x3", "hsmssd_bwd_gate", "hsmssd_bwd_passB")):   # one kernel per call
            _lib.check(_call((nm + tag, (B, C, Hs)), lib.kmu_hsmssd_bwd_stage_x3_pk, _ptr(xn), _ptr(dy), _ptr(dh), _ptr(w_bcdt), _ptr(w_dw),
                             _ptr(w_hz), _ptr(w_out), _ptr(D), _ptr(state), _ptr(dx), _ptr(p_bcdt), _ptr(p_dw), _ptr(p_hz), _ptr(p_out),
                             _ptr(p_D), _ptr(ws), nbytes, B, C, N, Hs, stage, groups, _ptr(wpk), st), "kmu_hsmssd_bwd_stage_x3_pk")
        return dx, (p_bcdt, p_dw, p_hz, p_out, p_D, None, None)
    Pc = lib.kmu_mixer_bwd_partials(B, C)
    pc_W, pc_dw = mk(Pc, N, C), mk(Pc, N, 9)          # the C-row sections [:, N:2N] of p_bcdt / p_dw stay unwritten: these carry them
    nbytes = lib.kmu_mixer_bwd_ws_bytes(B, C, N, Hs)
    ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
    wpk = _hsm_pack(pack_ok, w_bcdt, w_dw, C, st, groups)
    for stage, nm in enumerate(_MIXER_BWD_STAGES):                                   # one kernel per call
        _lib.check(_call((nm + tag, (B, C, Hs)), lib.kmu_mixer_bwd_stage, _ptr(xn), _ptr(dy), _ptr(dh), _ptr(w_bcdt), _ptr(w_dw), _ptr(w_hz),
                         _ptr(w_out), _ptr(D), _ptr(state), _ptr(dx), _ptr(p_bcdt), _ptr(p_dw), _ptr(p_hz), _ptr(p_out), _ptr(p_D), _ptr(pc_W),
                         _ptr(pc_dw), _ptr(ws), nbytes, B, C, N, Hs, stage, groups, _ptr(wpk), st), "kmu_mixer_bwd_stage")
    return dx, (p_bcdt, p_dw, p_hz, p_out, p_D, pc_W, pc_dw)


def _mixer_colsum_pairs(parts, outs, N, grp=lambda p: p):
    """(partial, result) pairs of one weight group for ops.colsum: parts from _mixer_backward, outs = (d_bcdt [3N, C], d_dw [3N, 9],
    d_hz, d_out, d_D); grp selects the group's rows of a partial array (grouped launches)."""
    p_bcdt, p_dw, p_hz, p_out, p_D, pc_W, pc_dw = parts
    d_bcdt, d_dw, d_hz, d_out, d_D = outs
    tail = [(grp(p_hz), d_hz), (grp(p_out), d_out), (grp(p_D.view(-1, 1)), d_D)]
    if pc_W is None:
        return [(grp(p_bcdt), d_bcdt), (grp(p_dw), d_dw)] + tail
    pb, pd = grp(p_bcdt), grp(p_dw)
    return [(pb[..., :N, :], d_bcdt[:N]), (pb[..., 2 * N:, :], d_bcdt[2 * N:]), (grp(pc_W), d_bcdt[N:2 * N]), (pd[..., :N, :], d_dw[:N]),
            (pd[..., 2 * N:, :], d_dw[2 * N:]), (grp(pc_dw), d_dw[N:2 * N])] + tail


def _mixer_colsums(parts, N, C, defer):
    """Column sums of _mixer_backward's partial arrays (one weight group) in one deferred launch -> (d_bcdt, d_dw, d_hz, d_out, d_D)."""
    dev = parts[0].device
    mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    outs = (mk(3 * N, C), mk(3 * N, 9), mk(2 * C, C), mk(C, C), mk(1))
    pairs = _mixer_colsum_pairs(parts, outs, N)
    _wgrad(lambda: colsum(*[p for p, _ in pairs], outs=[o for _, o in pairs]), defer)
    d_bcdt, d_dw, d_hz, d_out, d_D = outs
    return d_bcdt.view(3 * N, C, 1), d_dw.view(3 * N, 1, 3, 3), d_hz.view(2 * C, C, 1), d_out.view(C, C, 1), d_D.view(1)


class HsmssdFn(torch.autograd.Function):
    """(y[B,C,Hs,Hs], h[B,C,N]) = HSMSSD(x[B,C,L])  (efficient_vim_init.py:33-61), the round-2 / round-1 kernels:
    K2_MATH "bf16x3" (csrc/hsmssd_x3.inc) or "f32" (csrc/hsmssd.hip); the default forward is MixerFn (csrc/hsmssd_v2.inc)."""

    @staticmethod
    def forward(ctx, x, w_bcdt, w_dw, w_hz, w_out, A, D):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, L = x.shape
        Hs = int(round(L ** 0.5))
        if Hs * Hs != L:
            raise RuntimeError("HSMSSD: L=%d is not a perfect square (reference: int(math.sqrt(L)))" % L)
        N = A.shape[0]
        origs = (w_bcdt, w_dw)
        w_bcdt, w_dw = _f32c(w_bcdt, "BCdt_proj.weight").reshape(3 * N, C), _f32c(w_dw, "dw.weight").reshape(3 * N, 9)
        ctx.pack_ok = _pack_ok(origs, (w_bcdt, w_dw))
        w_hz, w_out, D = _f32c(w_hz, "hz_proj.weight").reshape(2 * C, C), _f32c(w_out, "out_proj.weight").reshape(C, C), _f32c(D, "D")
        dev = x.device
        y = torch.empty(B, C, Hs, Hs, device=dev, dtype=torch.float32)
        h = torch.empty(B, C, N, device=dev, dtype=torch.float32)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev, dtype=torch.float32)
        nbytes = lib.kmu_hsmssd_fwd_ws_bytes(B, C, N, Hs)
        ws = torch.empty(max(1, nbytes // 4), device=dev, dtype=torch.float32)
        st = _stream()
        x3 = K2_MATH != "f32"
        fn, tail = lib.kmu_hsmssd_fwd_stage, (st,)
        if x3:      # composite-weight pack: per step when the weights are Parameters inside a pack_scope, else made here
            fn, tail = lib.kmu_hsmssd_fwd_stage_x3_pk, (1, _ptr(_hsm_pack(ctx.pack_ok, w_bcdt, w_dw, C, st)), st)
        for stage, nm in enumerate(("hsmssd_fwd_pass1", "hsmssd_fwd_gate", "hsmssd_fwd_pass2")):   # one kernel per call
            _lib.check(_call((nm + ("_x3" if x3 and stage != 1 else ""), (B, C, Hs)), fn, _ptr(x), _ptr(w_bcdt), _ptr(w_dw), _ptr(w_hz),
                             _ptr(w_out), _ptr(D), _ptr(y), _ptr(h), _ptr(state), _ptr(ws), nbytes, B, C, N, Hs, stage, *tail),
                       "kmu_hsmssd_fwd_stage")
        ctx.save_for_backward(x, w_bcdt, w_dw, w_hz, w_out, D, state)
        ctx.set_materialize_grads(False)      # EfficientViMBlock drops h: no zero tensor for its gradient
        ctx.dims = (B, C, N, Hs)
        ctx.defer_wgrad = _leaf(w_bcdt, w_dw, w_hz, w_out, D)
        ctx.zero_A = _const_zeros(A)       # dL/dA == 0 exactly: a shared constant, cached here (before any graph capture)
        return y, h

    @staticmethod
    def backward(ctx, dy, dh):
        x, w_bcdt, w_dw, w_hz, w_out, D, state = ctx.saved_tensors
        dx, (d_bcdt, d_dw, d_hz, d_out, d_D) = _hsmssd_backward(x, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, ctx.dims, ctx.pack_ok,
                                                                ctx.defer_wgrad)
        # softmax_L(dt + A[n]) is shift invariant => dL/dA == 0 exactly (the reference's autograd
        # returns ~1e-7 rounding noise here)
        return dx, d_bcdt, d_dw, d_hz, d_out, ctx.zero_A, d_D


# ticket words of the last-arriver gate (csrc/hsmssd_v2.inc): zero-initialised once, left zero by every launch; each call takes the
# next B words of the ring, so launches in flight on different streams never share a counter
_TICKETS = {}


def _tickets(dev, n):
    ent = _TICKETS.get(dev.index)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("kmunet: the first mixer call of a process must not be inside a hipGraph capture (ticket ring allocation)")
        ent = _TICKETS[dev.index] = [torch.zeros(1 << 16, device=dev, dtype=torch.int32), 0]
    pool, cur = ent
    if cur + n > pool.numel():
        cur = 0
    ent[1] = cur + n
    return pool[cur:cur + n]


class MixerFn(torch.autograd.Function):
    """(y[B,C,Hs,Hs], h[B,C,N][, x']) = HSMSSD(LayerNorm1D(x))  -- vim_utils_init.py:50-59 + efficient_vim_init.py:33-61 in TWO launches
    (csrc/hsmssd_v2.inc: LayerNorm folded into both passes' staging, the gate stage run by the last-arriving workgroup of pass 1,
    pass 2 as a per-sample dense 3x3).  ln_weight None: no LayerNorm (x is the mixer's input as it is).  alias: a third output
    aliasing x whose gradient the LayerNorm backward kernel adds in (EfficientViMBlock blends y with the x it normalised)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, w_bcdt, w_dw, w_hz, w_out, A, D, alias=False):
        lib = _lib.load()
        xin = x
        x = _f32c(x, "x")
        B, C, L = x.shape
        Hs = int(round(L ** 0.5))
        if Hs * Hs != L:
            raise RuntimeError("HSMSSD: L=%d is not a perfect square (reference: int(math.sqrt(L)))" % L)
        N = A.shape[0]
        origs = (w_bcdt, w_dw)
        w_bcdt, w_dw = _f32c(w_bcdt, "BCdt_proj.weight").reshape(3 * N, C), _f32c(w_dw, "dw.weight").reshape(3 * N, 9)
        ctx.pack_ok = _pack_ok(origs, (w_bcdt, w_dw))
        w_hz, w_out, D = _f32c(w_hz, "hz_proj.weight").reshape(2 * C, C), _f32c(w_out, "out_proj.weight").reshape(C, C), _f32c(D, "D")
        dev = x.device
        ln = ln_w is not None
        need_bwd = any(ctx.needs_input_grad)
        lw = lb = xn = stats = None
        if ln:
            lw, lb = _f32c(ln_w, "norm.weight").reshape(-1), _f32c(ln_b, "norm.bias").reshape(-1)
            if need_bwd:          # the backward kernels read the normalised x; inference skips the store
                xn = torch.empty_like(x)
                stats = torch.empty(B, L, 2, device=dev, dtype=torch.float32)
        y = torch.empty(B, C, Hs, Hs, device=dev, dtype=torch.float32)
        h = torch.empty(B, C, N, device=dev, dtype=torch.float32)
        state = torch.empty(lib.kmu_hsmssd_state_elems(B, C, N), device=dev, dtype=torch.float32)
        nbytes = lib.kmu_mixer_fwd_ws_bytes(B, C, N, Hs)
        ws = torch.empty(max(1, (nbytes + 3) // 4), device=dev, dtype=torch.float32)
        st = _stream()
        wpk = _hsm_pack(ctx.pack_ok, w_bcdt, w_dw, C, st)
        tk = _tickets(dev, B)
        for stage, nm in enumerate(("hsmssd_fwd_pass1_v2", "hsmssd_fwd_pass2_v2")):
            _lib.check(_call((nm, (B, C, Hs)), lib.kmu_mixer_fwd_stage, _ptr(x), _ptr(lw), _ptr(lb), float(eps), _ptr(w_dw), _ptr(w_hz),
                             _ptr(w_out), _ptr(D), _ptr(wpk), _ptr(y), _ptr(h), _ptr(state), _ptr(xn), _ptr(stats), _ptr(ws), nbytes,
                             _ptr(tk), B, C, N, Hs, stage, 1, st), "kmu_mixer_fwd_stage")
        ctx.ln = ln
        if ln:
            ctx.save_for_backward(x, lw, stats, xn, w_bcdt, w_dw, w_hz, w_out, D, state)
            ctx.lnshape = ln_w.shape
            ctx.defer_ln = _leaf(ln_w, ln_b)
        else:
            ctx.save_for_backward(x, w_bcdt, w_dw, w_hz, w_out, D, state)
        ctx.set_materialize_grads(False)
        ctx.dims = (B, C, N, Hs)
        ctx.defer_wgrad = _leaf(w_bcdt, w_dw, w_hz, w_out, D)
        ctx.zero_A = _const_zeros(A)
        if alias:
            return y, h, xin.view_as(xin)
        return y, h

    @staticmethod
    def backward(ctx, dy, dh, dalias=None):
        lib = _lib.load()
        if ctx.ln:
            x, lw, stats, xn, w_bcdt, w_dw, w_hz, w_out, D, state = ctx.saved_tensors
        else:
            x, w_bcdt, w_dw, w_hz, w_out, D, state = ctx.saved_tensors
            xn = x
        B, C, N, Hs = ctx.dims
        if dy is None and dh is None:                 # only the alias was used
            return (dalias,) + (None,) * 10
        dxn, parts = _mixer_backward(xn, dy, dh, w_bcdt, w_dw, w_hz, w_out, D, state, ctx.dims, ctx.pack_ok)
        d_bcdt, d_dw, d_hz, d_out, d_D = _mixer_colsums(parts, N, C, ctx.defer_wgrad)
        if not ctx.ln:
            if dalias is not None:
                dxn = dxn + dalias
            return dxn, None, None, None, d_bcdt, d_dw, d_hz, d_out, ctx.zero_A, d_D, None
        L = Hs * Hs
        addend = None if dalias is None else _f32c(dalias, "grad of the alias")
        rows = lib.kmu_layernorm1d_partials(B, C, L)
        dx = torch.empty_like(x)
        dwp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        dbp = torch.empty(rows, C, device=x.device, dtype=torch.float32)
        _lib.check(_call(("layernorm1d_bwd", (B, C, L)), lib.kmu_layernorm1d_bwd_add, _ptr(x), _ptr(lw), _ptr(stats), _ptr(dxn), _ptr(addend),
                         _ptr(dx), _ptr(dwp), _ptr(dbp), B, C, L, 1, _stream()), "kmu_layernorm1d_bwd_add")
        dw, db = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        _wgrad(lambda: colsum(dwp, dbp, outs=[dw, db]), ctx.defer_ln)
        return dx, dw.view(ctx.lnshape), db.view(ctx.lnshape), None, d_bcdt, d_dw, d_hz, d_out, ctx.zero_A, d_D, None


def hsmssd(x, w_bcdt, w_dw, w_hz, w_out, A, D):
    if K2_MATH == "v2":
        return MixerFn.apply(x, None, None, 0.0, w_bcdt, w_dw, w_hz, w_out, A, D)
    return HsmssdFn.apply(x, w_bcdt, w_dw, w_hz, w_out, A, D)


def mixer_ln(x, ln_w, ln_b, eps, w_bcdt, w_dw, w_hz, w_out, A, D, alias=False):
    """HSMSSD(LayerNorm1D(x)) -> (y, h[, x']); K2_MATH "v2" (default): two launches, otherwise the LayerNorm kernel + the older forward."""
    if K2_MATH == "v2":
        if alias and not (torch.is_grad_enabled() and x.requires_grad):
            y, h = MixerFn.apply(x, ln_w, ln_b, eps, w_bcdt, w_dw, w_hz, w_out, A, D)
            return y, h, x
        return MixerFn.apply(x, ln_w, ln_b, eps, w_bcdt, w_dw, w_hz, w_out, A, D, alias)
    if alias:
        xn, xa = layernorm1d_alias(x, ln_w, ln_b, eps)
        return HsmssdFn.apply(xn, w_bcdt, w_dw, w_hz, w_out, A, D) + (xa,)
    return HsmssdFn.apply(layernorm1d(x, ln_w, ln_b, eps), w_bcdt, w_dw, w_hz, w_out, A, D)


# ------------------------------------------------------------------------------------------ K3
class DySampleFn(torch.autograd.Function):
    """DySample 'lp' sampling stage given the 1x1 offset-conv output (DySample_md.py:49-68)."""

    @staticmethod
    def forward(ctx, x, conv_out, init_pos, want_indices):
        lib = _lib.load()
        x, conv_out = _f32c(x, "x"), _f32c(conv_out, "offset conv output")
        ipos = _f32c(init_pos, "init_pos").reshape(-1)
        B, C, H, W = x.shape
        if conv_out.shape != (B, 32, H, W) or ipos.numel() != 32:
            raise RuntimeError("DySample: expected scale=2, groups=4 (32 offset channels), got %s" % (tuple(conv_out.shape),))
        y = torch.empty(B, C, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
        ix0 = iy0 = None
        if want_indices:
            ix0 = torch.empty(B * 4, 2 * H, 2 * W, device=x.device, dtype=torch.int32)
            iy0 = torch.empty_like(ix0)
        _lib.check(_call(("dysample_lp_fwd", (B, C, H, W)), lib.kmu_dysample_lp_fwd, _ptr(x), _ptr(conv_out), _ptr(ipos), _ptr(y),
                         _ptr(ix0), _ptr(iy0), B, C, H, W, _stream()), "kmu_dysample_lp_fwd")
        ctx.save_for_backward(x, conv_out, ipos)
        if want_indices:
            ctx.mark_non_differentiable(ix0, iy0)
            return y, ix0, iy0
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        lib = _lib.load()
        x, conv_out, ipos = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C, H, W = x.shape
        dx = torch.zeros_like(x)
        dconv = torch.empty_like(conv_out)
        _lib.check(_call(("dysample_lp_bwd", (B, C, H, W)), lib.kmu_dysample_lp_bwd, _ptr(x), _ptr(conv_out), _ptr(ipos), _ptr(dy),
                         _ptr(dx), _ptr(dconv), B, C, H, W, _stream()), "kmu_dysample_lp_bwd")
        return dx, dconv, None, None


def dysample_lp(x, conv_out, init_pos, return_indices=False):
    return DySampleFn.apply(x, conv_out, init_pos, return_indices)


# ------------------------------------------------------------------------------------------ K4
class DeformConv2dFn(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 deformable conv (torchvision.ops.DeformConv2d semantics, DAGEM_md.py:98-101)."""

    @staticmethod
    def forward(ctx, x, offset, weight, bias):
        lib = _lib.load()
        x, offset, weight = _f32c(x, "x"), _f32c(offset, "offset"), _f32c(weight, "weight")
        bias = _f32c(bias, "bias") if bias is not None else None
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        if tuple(weight.shape[1:]) != (Cin, 3, 3) or offset.shape != (B, 18, H, W):
            raise RuntimeError("DeformConv2d: only 3x3/stride1/pad1/one offset group is built (weight %s, offset %s)"
                               % (tuple(weight.shape), tuple(offset.shape)))
        y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
        _lib.check(lib.kmu_deform_conv2d_fwd(_ptr(x), _ptr(offset), _ptr(weight), _ptr(bias), _ptr(y), B, Cin, Cout, H, W,
                                             _stream()), "kmu_deform_conv2d_fwd")
        ctx.save_for_backward(x, offset, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, offset, weight = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dx, dw = torch.zeros_like(x), torch.zeros_like(weight)
        db = torch.zeros(Cout, device=x.device, dtype=torch.float32)
        doff = torch.empty_like(offset)
        _lib.check(lib.kmu_deform_conv2d_bwd(_ptr(x), _ptr(offset), _ptr(weight), _ptr(dy), _ptr(dx), _ptr(doff), _ptr(dw),
                                             _ptr(db), B, Cin, Cout, H, W, _stream()), "kmu_deform_conv2d_bwd")
        return dx, doff, dw, (db if ctx.has_bias else None)


def deform_conv2d_fused(x, offset, weight, bias=None):
    """Single-launch VALU variant (kmu_deform_conv2d_fwd/bwd): self-contained, but latency-bound at KM-UNet's size."""
    return DeformConv2dFn.apply(x, offset, weight, bias)


class DeformSampleFn(torch.autograd.Function):
    """cols [B, Cin*9, H*W] = bilinear samples of x at the 9 deformed taps of every output pixel (the im2col half of
    torchvision.ops.DeformConv2d, DAGEM_md.py:98-101); backward scatters d cols into dx and yields d offset."""

    @staticmethod
    def forward(ctx, x, offset):
        lib = _lib.load()
        x, offset = _f32c(x, "x"), _f32c(offset, "offset")
        B, Cin, H, W = x.shape
        cols = torch.empty(B, Cin * 9, H * W, device=x.device, dtype=torch.float32)
        _lib.check(_call(("deform_sample_fwd", (B, Cin, H, W)), lib.kmu_deform_sample_fwd, _ptr(x), _ptr(offset), _ptr(cols), B, Cin, H, W,
                         _stream()), "kmu_deform_sample_fwd")
        ctx.save_for_backward(x, offset)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        lib = _lib.load()
        x, offset = ctx.saved_tensors
        dcols = _f32c(dcols, "dcols")
        B, Cin, H, W = x.shape
        doff = torch.empty_like(offset)
        if lib.kmu_deform_sample_bwd_lds_supported(B, Cin, H, W):     # channel planes accumulated in LDS: no global atomics, no zero fill
            dx = torch.empty_like(x)
            _lib.check(_call(("deform_sample_bwd", (B, Cin, H, W)), lib.kmu_deform_sample_bwd_lds, _ptr(x), _ptr(offset), _ptr(dcols), _ptr(dx),
                             _ptr(doff), B, Cin, H, W, _stream()), "kmu_deform_sample_bwd_lds")
            return dx, doff
        dx = torch.zeros_like(x)
        _lib.check(_call(("deform_sample_bwd", (B, Cin, H, W)), lib.kmu_deform_sample_bwd, _ptr(x), _ptr(offset), _ptr(dcols), _ptr(dx),
                         _ptr(doff), B, Cin, H, W, _stream()), "kmu_deform_sample_bwd")
        return dx, doff


def deform_conv2d(x, offset, weight, bias=None):
    """Deformable 3x3 conv = sampling kernel (columns) + a [Cout, Cin*9] contraction on the pointwise-conv kernels (its two backward
    contractions replace 4.7 M float atomics of the fused kernel's weight gradient)."""
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (Cin, 3, 3) or tuple(offset.shape) != (B, 18, H, W):
        raise RuntimeError("DeformConv2d: only 3x3/stride1/pad1/one offset group is built (weight %s, offset %s)"
                           % (tuple(weight.shape), tuple(offset.shape)))
    cols = DeformSampleFn.apply(x, offset)
    if pwconv_supported(Cin * 9, Cout, H * W):
        # the [Cout, Cin*9] x [Cin*9, HW] product is a 1x1 convolution over the sampled columns: the pointwise-conv kernels (forward,
        # input gradient, weight + bias gradient), contraction tiled over 256 channels at a time -- no rocBLAS call on the model's path
        return pwconv(cols.view(B, Cin * 9, H, W), weight.reshape(Cout, Cin * 9), bias)
    y = torch.matmul(weight.reshape(Cout, Cin * 9), cols)      # shapes outside the kernels' tiling (tests: 8 channels, 7 x 9 pixels)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y.view(B, Cout, H, W)


# ------------------------------------------------------------------------------------------ depthwise 3x3
class DwConv3x3Fn(torch.autograd.Function):
    """Depthwise 3x3/s1/p1 conv (ConvLayer2D with groups=dim, vim_utils_init.py:62-89; DirectionAttention.conv,
    KM_UNetV3_SH.py:222)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.defer_wgrad = _leaf(weight, bias)
        lib = _lib.load()
        x, w = _f32c(x, "x"), _f32c(weight, "weight")
        b = _f32c(bias, "bias") if bias is not None else None
        B, C, H, W = x.shape
        if tuple(w.shape) != (C, 1, 3, 3):
            raise RuntimeError("dwconv3x3: weight %s does not match a depthwise 3x3 conv over %d channels" % (tuple(w.shape), C))
        y = torch.empty_like(x)
        _lib.check(_call(("dwconv3x3_fwd", (B, C, H, W)), lib.kmu_dwconv3x3_fwd, _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, C, H, W,
                         _stream()), "kmu_dwconv3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C, H, W = x.shape
        st = _stream()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(_call(("dwconv3x3_bwd_data", (B, C, H, W)), lib.kmu_dwconv3x3_bwd_data, _ptr(dy), _ptr(w), _ptr(dx), B, C, H,
                             W, st), "kmu_dwconv3x3_bwd_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(C, 1, 3, 3, device=x.device, dtype=torch.float32)
            db = torch.empty(C, device=x.device, dtype=torch.float32) if ctx.has_bias else None

            def job():
                P = lib.kmu_dwconv3x3_partials(B)
                dwp = torch.empty(P, C, 9, device=x.device, dtype=torch.float32)
                dbp = torch.empty(P, C, device=x.device, dtype=torch.float32) if db is not None else None
                _lib.check(_call(("dwconv3x3_bwd_weight", (B, C, H, W)), lib.kmu_dwconv3x3_bwd_weight, _ptr(x), _ptr(dy), _ptr(dwp),
                                 _ptr(dbp), B, C, H, W, _stream()), "kmu_dwconv3x3_bwd_weight")
                colsum(dwp, dbp, outs=[dw.view(C, 9)] + ([db] if db is not None else []))
            _wgrad(job, ctx.defer_wgrad)
        return dx, dw, db


def dwconv3x3(x, weight, bias=None):
    return DwConv3x3Fn.apply(x, weight, bias)


class DwConv3x3ScaledFn(torch.autograd.Function):
    """y = scale[b,c] * (depthwise3x3(x) + bias): DirectionAttention's conv(attn) * gate (KM_UNetV3_SH.py:262-263)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale):
        lib = _lib.load()
        x, w = _f32c(x, "x"), _f32c(weight, "weight")
        b = None if bias is None else _f32c(bias, "bias")
        B, C, H, W = x.shape
        sc = _f32c(scale, "scale").view(B, C)
        y = torch.empty_like(x)
        _lib.check(_call(("dwconv3x3_scaled_fwd", (B, C, H, W)), lib.kmu_dwconv3x3_scaled_fwd, _ptr(x), _ptr(w), _ptr(b), _ptr(sc), _ptr(y),
                         B, C, H, W, _stream()), "kmu_dwconv3x3_scaled_fwd")
        ctx.save_for_backward(x, w, b, sc)
        ctx.sshape = scale.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, b, sc = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C, H, W = x.shape
        st, dev = _stream(), x.device
        dx = None
        P = lib.kmu_dwconv3x3_partials(B)
        dwp = torch.empty(P, C, 9, device=dev, dtype=torch.float32)
        dbp = torch.empty(P, C, device=dev, dtype=torch.float32)
        if ctx.needs_input_grad[0] and DWBN_ALL and W % 4 == 0 and B <= 65535:      # data gradient and partials from one pass over dy
            dx = torch.empty_like(x)
            _lib.check(_call(("dwconv3x3_scaled_bwd_all", (B, C, H, W)), lib.kmu_dwconv3x3_scaled_bwd_all, _ptr(dy), _ptr(x), _ptr(w), _ptr(sc),
                             _ptr(dx), _ptr(dwp), _ptr(dbp), B, C, H, W, st), "kmu_dwconv3x3_scaled_bwd_all")
        else:
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(_call(("dwconv3x3_scaled_bwd_data", (B, C, H, W)), lib.kmu_dwconv3x3_scaled_bwd_data, _ptr(dy), _ptr(w), _ptr(sc),
                                 _ptr(dx), B, C, H, W, st), "kmu_dwconv3x3_scaled_bwd_data")
            _lib.check(_call(("dwconv3x3_bwd_weight", (B, C, H, W)), lib.kmu_dwconv3x3_bwd_weight, _ptr(x), _ptr(dy), _ptr(dwp), _ptr(dbp), B, C,
                             H, W, st), "kmu_dwconv3x3_bwd_weight")
        dw = torch.empty(C, 1, 3, 3, device=dev, dtype=torch.float32)
        db = torch.empty(C, device=dev, dtype=torch.float32) if b is not None else None
        ds = torch.empty(B, C, device=dev, dtype=torch.float32)
        _lib.check(_call(("dwconv3x3_scaled_finish", (B, C)), lib.kmu_dwconv3x3_scaled_finish, _ptr(dwp), _ptr(dbp), _ptr(sc), _ptr(w),
                         _ptr(b), _ptr(dw), _ptr(db), _ptr(ds), B, C, st), "kmu_dwconv3x3_scaled_finish")
        return dx, dw, db, ds.view(ctx.sshape)


def dwconv3x3_scaled(x, weight, bias, scale):
    return DwConv3x3ScaledFn.apply(x, weight, bias, scale)


class QkvGateDwFn(torch.autograd.Function):
    """y = scale[b,c] * (depthwise3x3(sigmoid(q*k)*v) + bias) on the packed qkv tensor: DirectionAttention's local gate folded into its
    stencil (KM_UNetV3_SH.py:258-263); attn is formed on the fly in both directions and never stored."""

    @staticmethod
    def forward(ctx, qkv, weight, bias, scale):
        lib = _lib.load()
        qkv, w = _f32c(qkv, "qkv"), _f32c(weight, "weight")
        b = None if bias is None else _f32c(bias, "bias")
        B, C3, H, W = qkv.shape
        C = C3 // 3
        sc = _f32c(scale, "scale").view(B, C)
        y = torch.empty(B, C, H, W, device=qkv.device, dtype=torch.float32)
        _lib.check(_call(("qkv_dw_scaled_fwd", (B, C, H, W)), lib.kmu_qkv_dw_scaled_fwd, _ptr(qkv), _ptr(w), _ptr(b), _ptr(sc), _ptr(y), B, C,
                         H, W, _stream()), "kmu_qkv_dw_scaled_fwd")
        ctx.save_for_backward(qkv, w, b, sc)
        ctx.sshape = scale.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        qkv, w, b, sc = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C3, H, W = qkv.shape
        C = C3 // 3
        st, dev = _stream(), qkv.device
        P = lib.kmu_dwconv3x3_partials(B)
        dwp = torch.empty(P, C, 9, device=dev, dtype=torch.float32)
        dbp = torch.empty(P, C, device=dev, dtype=torch.float32)
        dqkv = torch.empty_like(qkv)
        _lib.check(_call(("qkv_dw_scaled_bwd", (B, C, H, W)), lib.kmu_qkv_dw_scaled_bwd, _ptr(dy), _ptr(qkv), _ptr(w), _ptr(sc), _ptr(dqkv),
                         _ptr(dwp), _ptr(dbp), B, C, H, W, st), "kmu_qkv_dw_scaled_bwd")
        dw = torch.empty(C, 1, 3, 3, device=dev, dtype=torch.float32)
        db = torch.empty(C, device=dev, dtype=torch.float32) if b is not None else None
        ds = torch.empty(B, C, device=dev, dtype=torch.float32)
        _lib.check(_call(("dwconv3x3_scaled_finish", (B, C)), lib.kmu_dwconv3x3_scaled_finish, _ptr(dwp), _ptr(dbp), _ptr(sc), _ptr(w),
                         _ptr(b), _ptr(dw), _ptr(db), _ptr(ds), B, C, st), "kmu_dwconv3x3_scaled_finish")
        return dqkv, dw, db, ds.view(ctx.sshape)


def qkv_gate_dw_supported(B, C, H, W):
    return bool(_lib.load().kmu_qkv_dw_scaled_supported(B, C, H, W))


def qkv_gate_dw(qkv, weight, bias, scale):
    return QkvGateDwFn.apply(qkv, weight, bias, scale)


# ------------------------------------------------------------------------------------------ 3-tap axis convolutions
class Shift3Fn(torch.autograd.Function):
    """[B,C,H,W] -> [B,3C,H,W]: the three copies of x shifted by -1, 0, +1 along `axis` (0 = H, 1 = W), zero padded."""

    @staticmethod
    def forward(ctx, x, axis):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, H, W = x.shape
        out = torch.empty(B, 3 * C, H, W, device=x.device, dtype=torch.float32)
        _lib.check(_call(("shift3_fwd", (B, C, H, W)), lib.kmu_shift3_fwd, _ptr(x), _ptr(out), B, C, H, W, axis, _stream()), "kmu_shift3_fwd")
        ctx.cfg = (B, C, H, W, axis)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, C, H, W, axis = ctx.cfg
        g = _f32c(g, "grad")
        dx = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32)
        _lib.check(_call(("shift3_bwd", (B, C, H, W)), lib.kmu_shift3_bwd, _ptr(g), _ptr(dx), B, C, H, W, axis, _stream()), "kmu_shift3_bwd")
        return dx, None


def conv3tap(x, weight, bias, axis):
    """nn.Conv2d with a (3,1) (axis 0) or (1,3) (axis 1) kernel, stride 1, padding 1 on that axis (KM_UNetV3_SH.py:170-172):
    tap stacking + the pointwise-conv kernels."""
    # channel-major tap order: the stacked input's channel 3 ci + t pairs with W[co, ci, t] -- the weight tensor itself, read as
    # [Co, 3 Ci] inside PwConvFn (no permuted copy, the Parameter stays a leaf: its gradient is deferred like any other)
    return pwconv(Shift3Fn.apply(x, axis), weight, bias)


# ------------------------------------------------------------------------------------------ branch fusion
class Mix3Fn(torch.autograd.Function):
    """out = x + s[b] * (g[b,0] f0 + g[b,1] f1 + g[b,2] f2)  (KM_UNetV3_SH.py:141-146); s = DropPath scale or None."""

    @staticmethod
    def forward(ctx, x, f0, f1, f2, g, s):
        lib = _lib.load()
        x, f0, f1, f2 = _f32c(x, "x"), _f32c(f0, "f0"), _f32c(f1, "f1"), _f32c(f2, "f2")
        B = x.shape[0]
        n = x.numel() // B
        gc = _f32c(g, "g").view(B, 3)
        sc = None if s is None else _f32c(s, "s").view(B)
        out = torch.empty_like(x)
        _lib.check(_call(("mix3_fwd", (B, n)), lib.kmu_mix3_fwd, _ptr(x), _ptr(f0), _ptr(f1), _ptr(f2), _ptr(gc), _ptr(sc), _ptr(out), B, n,
                         _stream()), "kmu_mix3_fwd")
        ctx.save_for_backward(f0, f1, f2, gc, sc)
        ctx.gshape = g.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        f0, f1, f2, gc, sc = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B = dy.shape[0]
        n = dy.numel() // B
        d0, d1, d2 = torch.empty_like(f0), torch.empty_like(f1), torch.empty_like(f2)
        part = torch.empty(lib.kmu_mix3_blocks(n), B * 3, device=dy.device, dtype=torch.float32)
        _lib.check(_call(("mix3_bwd", (B, n)), lib.kmu_mix3_bwd, _ptr(dy), _ptr(f0), _ptr(f1), _ptr(f2), _ptr(gc), _ptr(sc), _ptr(d0),
                         _ptr(d1), _ptr(d2), _ptr(part), B, n, _stream()), "kmu_mix3_bwd")
        (dg,) = colsum(part)
        return dy, d0, d1, d2, dg.view(ctx.gshape), None


def mix3(x, f0, f1, f2, g, s=None):
    return Mix3Fn.apply(x, f0, f1, f2, g, s)


class SpatialMeanFn(torch.autograd.Function):
    """x.mean(dim=(2, 3)) of [B,C,H,W] -> [B,C] (nn.AdaptiveAvgPool2d(1) of the squeeze-excite gates: KM_UNetV3_SH.py:231, :320,
    :342).  One deterministic launch; the backward returns the broadcast (grad / HW) as an EXPANDED view -- autograd's fan-in add
    consumes it without the full-size division ATen's mean backward launches."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        out = torch.empty(B, C, device=x.device, dtype=torch.float32)
        _lib.check(_call(("mean_rows", (B, C, HW)), lib.kmu_mean_rows, _ptr(x), None, None, _ptr(out), B, C, HW, 1, _stream()), "kmu_mean_rows")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C = ctx.shape[:2]
        HW = 1
        for d in ctx.shape[2:]:
            HW *= d
        return (g * (1.0 / HW)).view(B, C, *([1] * (len(ctx.shape) - 2))).expand(ctx.shape)


def spatial_mean(x):
    return SpatialMeanFn.apply(x)


class GatedMix3Fn(torch.autograd.Function):
    """EnhancedViMBlock's fusion gate + branch mix as ONE autograd node (KM_UNetV3_SH.py:111-117, :141-146):
        g   = softmax(W2 gelu(W1 mean_hw(cat(f0, f1, f2)) + b1) + b2)          [B,3]
        out = x + s[b] (g0 f0 + g1 f1 + g2 f2)
    forward 3 launches (pool, gate MLP, mix), backward 4 (d g partials, their column sum, gate MLP backward, d f_t = s g_t dy +
    d pooled / HW) -- the separate nodes took 3 means + cat forward and 3 full-size divisions + 3 fan-in adds backward on top."""

    @staticmethod
    def forward(ctx, x, f0, f1, f2, w1, b1, w2, b2, s):
        lib = _lib.load()
        x, f0, f1, f2 = _f32c(x, "x"), _f32c(f0, "f0"), _f32c(f1, "f1"), _f32c(f2, "f2")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        n = C * HW
        dev, st = x.device, _stream()
        Hd = w1.shape[0]
        w1c, w2c = _f32c(w1, "w1").view(Hd, 3 * C), _f32c(w2, "w2").view(3, Hd)
        b1c, b2c = _f32c(b1, "b1"), _f32c(b2, "b2")
        pooled = torch.empty(B, 3 * C, device=dev, dtype=torch.float32)
        _lib.check(_call(("mean_rows", (B, 3 * C, HW)), lib.kmu_mean_rows, _ptr(f0), _ptr(f1), _ptr(f2), _ptr(pooled), B, C, HW, 3, st),
                   "kmu_mean_rows")
        z1 = torch.empty(B, Hd, device=dev, dtype=torch.float32)
        g = torch.empty(B, 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("gate_mlp_fwd", (B, 3 * C, Hd, 3)), lib.kmu_gate_mlp_fwd, _ptr(pooled), _ptr(w1c), _ptr(b1c), _ptr(w2c), _ptr(b2c),
                         _ptr(z1), _ptr(g), B, 3 * C, Hd, 3, _ACT1["gelu"], _ACT2["softmax"], st), "kmu_gate_mlp_fwd")
        sc = None if s is None else _f32c(s, "s").view(B)
        out = torch.empty_like(x)
        _lib.check(_call(("mix3_fwd", (B, n)), lib.kmu_mix3_fwd, _ptr(x), _ptr(f0), _ptr(f1), _ptr(f2), _ptr(g), _ptr(sc), _ptr(out), B, n, st),
                   "kmu_mix3_fwd")
        ctx.save_for_backward(f0, f1, f2, g, sc, pooled, w1c, w2c, z1)
        ctx.cfg = (B, C, HW, Hd, tuple(w1.shape), tuple(w2.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        f0, f1, f2, g, sc, pooled, w1c, w2c, z1 = ctx.saved_tensors
        B, C, HW, Hd, s1, s2 = ctx.cfg
        n = C * HW
        dy = _f32c(dy, "dy")
        dev, st = dy.device, _stream()
        part = torch.empty(lib.kmu_mix3_blocks(n), B * 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("mix3_bwd_dg", (B, n)), lib.kmu_mix3_bwd_dg, _ptr(dy), _ptr(f0), _ptr(f1), _ptr(f2), _ptr(g), _ptr(sc), _ptr(part),
                         B, n, st), "kmu_mix3_bwd_dg")
        (dg,) = colsum(part)
        dpool = torch.empty_like(pooled)
        dw1, dw2 = torch.empty(Hd, 3 * C, device=dev), torch.empty(3, Hd, device=dev)
        db1, db2 = torch.empty(Hd, device=dev), torch.empty(3, device=dev)
        _lib.check(_call(("gate_mlp_bwd", (B, 3 * C, Hd, 3)), lib.kmu_gate_mlp_bwd, _ptr(pooled), _ptr(w1c), _ptr(w2c), _ptr(z1), _ptr(g),
                         _ptr(dg), _ptr(dpool), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), B, 3 * C, Hd, 3, _ACT1["gelu"], _ACT2["softmax"],
                         st), "kmu_gate_mlp_bwd")
        d0, d1, d2 = torch.empty_like(f0), torch.empty_like(f1), torch.empty_like(f2)
        _lib.check(_call(("mix3_bwd_apply", (B, n)), lib.kmu_mix3_bwd_apply, _ptr(dy), _ptr(g), _ptr(sc), _ptr(dpool), _ptr(d0), _ptr(d1),
                         _ptr(d2), B, C, HW, st), "kmu_mix3_bwd_apply")
        return dy, d0, d1, d2, dw1.view(s1), db1, dw2.view(s2), db2, None


def gated_mix3(x, f0, f1, f2, lin1, lin2, s=None):
    return GatedMix3Fn.apply(x, f0, f1, f2, lin1.weight, lin1.bias, lin2.weight, lin2.bias, s)


# ------------------------------------------------------------------------------------------ LocalContrastAttention output
class LcaApplyFn(torch.autograd.Function):
    """x * (1 - g) + g with g [B, C] (KM_UNetV3_SH.py:366-368, torch.lerp(x, 1, g)): one launch each way."""

    @staticmethod
    def forward(ctx, x, g):
        lib = _lib.load()
        x, g = _f32c(x, "x"), _f32c(g, "gate")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        y = torch.empty_like(x)
        _lib.check(_call(("lca_fwd", (B, C, HW)), lib.kmu_lca_fwd, _ptr(x), _ptr(g), _ptr(y), B, C, HW, _stream()), "kmu_lca_fwd")
        ctx.save_for_backward(x, g)
        ctx.gshape = tuple(g.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, g = ctx.saved_tensors
        dy = _f32c(dy, "dy")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        dx = torch.empty_like(x)
        dg = torch.empty(B, C, device=x.device, dtype=torch.float32)
        _lib.check(_call(("lca_bwd", (B, C, HW)), lib.kmu_lca_bwd, _ptr(x), _ptr(g), _ptr(dy), _ptr(dx), _ptr(dg), B, C, HW, _stream()),
                   "kmu_lca_bwd")
        return dx, dg.view(ctx.gshape)


def lca_apply(x, g):
    return LcaApplyFn.apply(x, g)


# ------------------------------------------------------------------------------------------ DAGEM edge features
class DagemEdgesFn(torch.autograd.Function):
    """edge[b,c,h,w,k] = x * roll_k(x), k = roll(+1, H), roll(-1, H), roll(+1, W), roll(-1, W)  (DAGEM_md.py:56-62)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, H, W = x.shape
        edge = torch.empty(B, C, H, W, 4, device=x.device, dtype=torch.float32)
        _lib.check(_call(("dagem_edges_fwd", (B, C, H, W)), lib.kmu_dagem_edges_fwd, _ptr(x), _ptr(edge), B, C, H, W, _stream()),
                   "kmu_dagem_edges_fwd")
        ctx.save_for_backward(x)
        return edge

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        _lib.check(_call(("dagem_edges_bwd", (B, C, H, W)), lib.kmu_dagem_edges_bwd, _ptr(x), _ptr(g), _ptr(dx), B, C, H, W, _stream()),
                   "kmu_dagem_edges_bwd")
        return dx


class DagemFn(torch.autograd.Function):
    """DAGEM_md.py:56-111 minus the deformable convolution, one launch per BatchNorm boundary (csrc/dagem_fused.hip): 4 launches
    forward, 5 + one column-sum launch backward.  forward(x, dconv, bns, training, wa, ba, wv, bv, we, be, wr, br, wf, gamma x 5,
    beta x 5) -> out; dconv = deform_conv(x, offset_conv(x)) WITHOUT the residual (added in stage 2).  bns: the five BatchNorm modules in
    the order edge_aggregation, vertex_update, edge_update, update_edge_reduce, final (running statistics updated in place)."""

    NP = 19

    @staticmethod
    def _args(x, dconv, bns, training, params):
        B, C, H, W = x.shape
        a = _lib.DagemArgs()
        a.B, a.C, a.H, a.W, a.training = B, C, H, W, int(training)
        for i, bn in enumerate(bns):
            a.eps[i], a.momentum[i] = float(bn.eps), float(bn.momentum)
            a.gamma[i], a.beta[i] = _ptr(params[9 + i]), _ptr(params[14 + i])
            a.running_mean[i], a.running_var[i] = _ptr(bn.running_mean), _ptr(bn.running_var)
            a.num_batches_tracked[i] = _ptr(bn.num_batches_tracked) if training else None
        a.x, a.dconv = _ptr(x), _ptr(dconv)
        for name, t in zip(("wa", "ba", "wv", "bv", "we", "be", "wr", "br", "wf"), params[:9]):
            setattr(a, name, _ptr(t))
        return a

    @staticmethod
    def forward(ctx, x, dconv, bns, training, *params):
        import ctypes
        lib = _lib.load()
        x, dconv = _f32c(x, "x"), _f32c(dconv, "deform_conv output")
        params = tuple(_f32c(p, "DAGEM parameter") for p in params)
        B, C, H, W = x.shape
        P, C2, dev = H * W, C // 2, x.device
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        sv = dict(a_pre=mk(B, C, P), u_pre=mk(B, C2, P, 4), v_pre=mk(B, C2, P), r_pre=mk(B, C2, P), z=mk(B, C, P), bnstat=mk(5, C, 2),
                  part=mk(lib.kmu_dagem_part_floats(B, C, H, W)))
        out = mk(B, C, H, W)
        a = DagemFn._args(x, dconv, bns, training, params)
        for k, t in sv.items():
            setattr(a, k, _ptr(t))
        a.out = _ptr(out)
        tap = None
        if RELU_TAP is not None:
            tap = dict(agg_out=mk(B, C, P), u_out=mk(B, C2, P, 4), vert_out=mk(B, C2, P), ue_out=mk(B, C2, P))
            for k, t in tap.items():
                setattr(a, k, _ptr(t))
        st = _stream()
        for stage in range(4):
            _lib.check(_call(("dagem_fwd%d" % stage, (B, C, H, W)), lib.kmu_dagem_stage, ctypes.byref(a), stage, st), "kmu_dagem_stage")
        if tap is not None:      # the five ReLU masks in the reference's call order and row layouts (DAGEM_md.py:65-104)
            RELU_TAP.append((tap["agg_out"] > 0).reshape(-1, 1).cpu())
            RELU_TAP.append((tap["vert_out"] > 0).permute(0, 2, 1).reshape(-1, C2).cpu())
            RELU_TAP.append((tap["u_out"] > 0).permute(0, 2, 3, 1).reshape(-1, C2).cpu())
            RELU_TAP.append((tap["ue_out"] > 0).reshape(-1, 1).cpu())
            RELU_TAP.append((out.detach() > 0).cpu())
        ctx.save_for_backward(x, dconv, *params, *sv.values())
        ctx.sv_keys = tuple(sv)
        ctx.bns, ctx.training = bns, training
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.defer = _leaf(*params)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        lib = _lib.load()
        t = ctx.saved_tensors
        x, dconv, params = t[0], t[1], t[2:2 + DagemFn.NP]
        sv = dict(zip(ctx.sv_keys, t[2 + DagemFn.NP:]))
        g = _f32c(g, "grad")
        B, C, H, W = x.shape
        P, C2, dev = H * W, C // 2, x.device
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        a = DagemFn._args(x, dconv, ctx.bns, ctx.training, params)
        for k, v in sv.items():
            setattr(a, k, _ptr(v))
        NW = lib.kmu_dagem_tiles(B, H, W)
        sc = dict(part_bwd=mk(lib.kmu_dagem_part_floats(B, C, H, W)), g_dd=mk(B, C, H, W), gv=mk(B, C2, P), gr=mk(B, C2, P), ga=mk(B, C, P),
                  dr_pre=mk(B, C2, P), dxb=mk(B, C, P), de=mk(B, C, P, 4), dx=mk(B, C, H, W), p_wf=mk(NW, C, C + C2), p_wv=mk(NW, C2, 2 * C),
                  p_bv=mk(NW, C2), p_we=mk(NW, C2, 2 * C), p_be=mk(NW, C2), p_wa=mk(NW, 5), p_wr=mk(NW, 5))
        for k, v in sc.items():
            setattr(a, k, _ptr(v))
        a.g_out = _ptr(g)
        dg, db = [mk(*ctx.shapes[9 + i]) for i in range(5)], [mk(*ctx.shapes[14 + i]) for i in range(5)]
        for i in range(5):
            a.d_gamma[i], a.d_beta[i] = _ptr(dg[i]), _ptr(db[i])
        st = _stream()
        for stage in range(4, 9):
            _lib.check(_call(("dagem_bwd%d" % (stage - 4), (B, C, H, W)), lib.kmu_dagem_stage, ctypes.byref(a), stage, st), "kmu_dagem_stage")
        dwa, dba, dwv, dbv, dwe, dbe, dwr, dbr, dwf = [mk(*ctx.shapes[i]) for i in range(9)]
        pa, pr = sc["p_wa"], sc["p_wr"]
        _wgrad(lambda: colsum(sc["p_wf"], sc["p_wv"], sc["p_bv"], sc["p_we"], sc["p_be"], pa[:, :4], pa[:, 4:], pr[:, :4], pr[:, 4:],
                              outs=[dwf, dwv, dbv, dwe, dbe, dwa, dba, dwr, dbr]), ctx.defer)
        return (sc["dx"], sc["g_dd"], None, None, dwa, dba, dwv, dbv, dwe, dbe, dwr, dbr, dwf, *dg, *db)


def dagem_supported(x, bns):
    return (x.is_cuda and x.dim() == 4 and bool(_lib.load().kmu_dagem_supported(x.shape[1]))
            and all(bn.momentum is not None and bn.track_running_stats and bn.affine for bn in bns))


def dagem_glue(x, dconv, mod):
    """Everything of DAGEM.forward (DAGEM_md.py:56-111) around the deformable convolution: out = final(cat(dconv + x, vertex * edge))."""
    seqs = (mod.edge_aggregation_func, mod.vertex_update_func, mod.edge_update_func, mod.update_edge_reduce_func)
    bns = [s_[1] for s_ in seqs] + [mod.final_aggregation_layer[1]]
    lin = {k: s_[0] for k, s_ in zip("aver", seqs)}
    params = (lin["a"].weight, lin["a"].bias, lin["v"].weight, lin["v"].bias, lin["e"].weight, lin["e"].bias, lin["r"].weight, lin["r"].bias,
              mod.final_aggregation_layer[0].weight, *[bn.weight for bn in bns], *[bn.bias for bn in bns])
    return DagemFn.apply(x, dconv, bns, mod.training, *params)


def dagem_edges(x):
    return DagemEdgesFn.apply(x)


# ------------------------------------------------------------------------------------------ EnhancedViMBlock tail
def _k_tn_fwd(lib, x, gh, bh, gw, bw, gc, bc, eps_gn, eps_ln):
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    stats = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(B * C * lib.kmu_triple_norm_splits(HW) * 2, device=x.device, dtype=torch.float32)
    _lib.check(_call(("triple_norm_fwd", (B, C, HW)), lib.kmu_triple_norm_fwd, _ptr(x), _ptr(gh), _ptr(bh), _ptr(gw), _ptr(bw), _ptr(gc),
                     _ptr(bc), _ptr(y), _ptr(stats), _ptr(ws), B, C, HW, float(eps_gn), float(eps_ln), _stream()), "kmu_triple_norm_fwd")
    return y, stats


def _k_tn_bwd(lib, x, dy, gh, gw, gc, stats, addend, eps_ln):
    B, C = x.shape[:2]
    HW = x.numel() // (B * C)
    dev = x.device
    dx = torch.empty_like(x)
    dgp, dbp = torch.empty(B, C, device=dev, dtype=torch.float32), torch.empty(B, C, device=dev, dtype=torch.float32)
    dcp = torch.empty(lib.kmu_triple_norm_partials(B, C, HW), C, device=dev, dtype=torch.float32)
    ws = torch.empty(B * C * lib.kmu_triple_norm_splits(HW) * 2, device=dev, dtype=torch.float32)
    _lib.check(_call(("triple_norm_bwd", (B, C, HW)), lib.kmu_triple_norm_bwd, _ptr(x), _ptr(dy), _ptr(gh), _ptr(gw), _ptr(gc), _ptr(stats),
                     _ptr(addend), _ptr(dx), _ptr(dgp), _ptr(dbp), _ptr(dcp), _ptr(ws), B, C, HW, float(eps_ln), _stream()),
               "kmu_triple_norm_bwd")
    return dx, dgp, dbp, dcp


def triple_norm_supported(C, HW):
    return bool(_lib.load().kmu_triple_norm_supported(C, HW))


class TripleNormFn(torch.autograd.Function):
    """(GroupNorm_h(x) + GroupNorm_w(x) + LayerNorm_c(x)) / 3 (KM_UNetV3_SH.py:266-284) as one node: 2 launches each way."""

    @staticmethod
    def forward(ctx, x, gh, bh, gw, bw, gc, bc, eps_gn, eps_ln):
        lib = _lib.load()
        ctx.defer_wgrad = _leaf(gh, bh, gw, bw, gc, bc)
        x = _f32c(x, "x")
        p = [_f32c(t, "norm parameter") for t in (gh, bh, gw, bw, gc, bc)]
        y, stats = _k_tn_fwd(lib, x, *p, eps_gn, eps_ln)
        ctx.save_for_backward(x, p[0], p[2], p[4], stats)
        ctx.eps_ln = float(eps_ln)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gh, gw, gc, stats = ctx.saved_tensors
        dx, dgp, dbp, dcp = _k_tn_bwd(lib, x, _f32c(dy, "dy"), gh, gw, gc, stats, None, ctx.eps_ln)
        C = x.shape[1]
        dG, dB, dC = (torch.empty(C, device=x.device, dtype=torch.float32) for _ in range(3))
        _wgrad(lambda: colsum(dgp, dbp, dcp, outs=[dG, dB, dC]), ctx.defer_wgrad)
        return dx, dG, dB, dG, dB, dC, dB, None, None


class VimTailFn(torch.autograd.Function):
    """EnhancedViMBlock's tail as ONE autograd node (KM_UNetV3_SH.py:147-150):
        out = x + s[b] * ffn2(GELU(ffn0(TripleNorm(x))))        s = DropPath's per-sample factor or None
    forward 4 launches (TripleNorm sums + apply, two 1x1 convs with GELU-on-load, bias, residual and s folded in), backward 4 on
    the dependent chain (two input gradients with GELU' / s folded in, TripleNorm sums + apply with the residual gradient as addend);
    the separate nodes took 11 and ~22 -- on the main stream, where nothing else runs meanwhile."""

    @staticmethod
    def forward(ctx, x, gh, bh, gw, bw, gc, bc, eps_gn, eps_ln, w0, b0, w2, b2, s):
        lib = _lib.load()
        ctx.defer_wgrad = _leaf(gh, bh, gw, bw, gc, bc, w0, b0, w2, b2)
        x = _f32c(x, "x")
        p = [_f32c(t, "norm parameter") for t in (gh, bh, gw, bw, gc, bc)]
        B, C, H, W = x.shape
        P, hid = H * W, w0.shape[0]
        w0c, w2c = _f32c(w0, "ffn[0].weight").view(hid, C), _f32c(w2, "ffn[2].weight").view(C, hid)
        b0c, b2c = _f32c(b0, "ffn[0].bias"), _f32c(b2, "ffn[2].bias")
        sc = None if s is None else _f32c(s, "s").view(B)
        st = _stream()
        n, stats = _k_tn_fwd(lib, x, *p, eps_gn, eps_ln)
        out = torch.empty_like(x)
        fused = TAIL_FUSED and hid == 4 * C and bool(lib.kmu_ffn_fused_supported(C, hid, P))
        if fused:       # fc1 -> GELU -> fc2 -> DropPath factor -> residual in ONE launch, the 4C-wide hidden tensor stays in registers
            _lib.check(_call(("tail_ffn_fwd", (B, C, P)), lib.kmu_tail_ffn_fwd, _ptr(n), _ptr(x), _ptr(w0c), _ptr(b0c), _ptr(w2c), _ptr(b2c),
                             _ptr(sc), _ptr(out), B, C, P, st), "kmu_tail_ffn_fwd")
            h = None
        else:
            h = torch.empty(B, hid, H, W, device=x.device, dtype=torch.float32)
            _lib.check(_call(("pwconv_fwd", (B, C, hid, P)), lib.kmu_pwconv_fwd, _ptr(n), _ptr(w0c), _ptr(b0c), _ptr(h), B, C, hid, P, 0, st),
                       "kmu_pwconv_fwd")
            _lib.check(_call(("pwconv_fwd_res", (B, hid, C, P)), lib.kmu_pwconv_fwd_res, _ptr(h), _ptr(w2c), _ptr(b2c), _ptr(x), _ptr(sc),
                             _ptr(out), B, hid, C, P, 1, st), "kmu_pwconv_fwd_res")
        ctx.save_for_backward(x, p[0], p[2], p[4], stats, n, h, w0c, w2c, sc, b0c)
        ctx.cfg = (float(eps_ln), tuple(w0.shape), tuple(w2.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, gh, gw, gc, stats, n, h, w0c, w2c, sc, b0c = ctx.saved_tensors
        eps_ln, s0, s2 = ctx.cfg
        g = _f32c(g, "grad")
        B, C, H, W = x.shape
        P, hid, dev, st = H * W, w0c.shape[0], x.device, _stream()
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        dw0, db0, dw2, db2, dG, dB, dC = mk(hid, C), mk(hid), mk(C, hid), mk(C), mk(C), mk(C), mk(C)
        if h is None:   # recompute form: one launch gives dn and the partials of all four parameter gradients
            rows = lib.kmu_ffn_fused_rows(B, C, P, 1)
            dn, slab0, slab2, rows_b = torch.empty_like(n), mk(rows, hid * C), mk(rows, C * hid), mk(rows, hid + C)
            _lib.check(_call(("tail_ffn_bwd", (B, C, P)), lib.kmu_tail_ffn_bwd, _ptr(n), _ptr(g), _ptr(w0c), _ptr(b0c), _ptr(w2c), _ptr(sc),
                             _ptr(dn), _ptr(slab0), _ptr(slab2), _ptr(rows_b), B, C, P, st), "kmu_tail_ffn_bwd")
            dx, dgp, dbp, dcp = _k_tn_bwd(lib, x, dn, gh, gw, gc, stats, g, eps_ln)
            dbb = mk(hid + C)
            _wgrad(lambda: colsum(slab0, slab2, rows_b, dgp, dbp, dcp, outs=[dw0.view(-1), dw2.view(-1), dbb, dG, dB, dC]), ctx.defer_wgrad)
            return dx, dG, dB, dG, dB, dC, dB, None, None, dw0.view(s0), dbb[:hid], dw2.view(s2), dbb[hid:], None
        dh = torch.empty_like(h)                    # d loss / d ffn0 output = s (W2^T g) GELU'(h)
        _lib.check(_call(("pwconv_bwd_input_s", (B, hid, C, P)), lib.kmu_pwconv_bwd_input_s, _ptr(g), _ptr(w2c), _ptr(h), _ptr(sc), _ptr(dh),
                         B, hid, C, P, 1, st), "kmu_pwconv_bwd_input_s")
        dn = torch.empty_like(n)
        _lib.check(_call(("pwconv_bwd_input", (B, C, hid, P)), lib.kmu_pwconv_bwd_input, _ptr(dh), _ptr(w0c), None, _ptr(dn), B, C, hid, P, 0,
                         st), "kmu_pwconv_bwd_input")
        dx, dgp, dbp, dcp = _k_tn_bwd(lib, x, dn, gh, gw, gc, stats, g, eps_ln)

        def job():
            sg = g if sc is None else g * sc.view(B, 1, 1, 1)
            _pw_wgrad_call(lib, h, sg, dw2, db2, B, hid, C, P, 1)
            _pw_wgrad_call(lib, n, dh, dw0, db0, B, C, hid, P, 0)
            colsum(dgp, dbp, dcp, outs=[dG, dB, dC])
        _wgrad(job, ctx.defer_wgrad)
        return dx, dG, dB, dG, dB, dC, dB, None, None, dw0.view(s0), db0, dw2.view(s2), db2, None


def vim_tail(x, norm, ffn0, ffn2, s=None):
    nh, nw, nc = norm.norm_h, norm.norm_w, norm.norm_c
    return VimTailFn.apply(x, nh.weight, nh.bias, nw.weight, nw.bias, nc.weight, nc.bias, nh.eps, nc.eps, ffn0.weight, ffn0.bias,
                           ffn2.weight, ffn2.bias, s)


# ------------------------------------------------------------------------------------------ SSIM window filter
class Gauss11Fn(torch.autograd.Function):
    """'valid' separable 11-tap filter over the last two dims (HybridLoss's SSIM window); backward = its adjoint."""

    @staticmethod
    def forward(ctx, x, taps):
        lib = _lib.load()
        x = _f32c(x, "x")
        H, W = x.shape[-2:]
        if H < 11 or W < 11:
            raise RuntimeError("gauss11: %dx%d is smaller than the 11x11 window" % (H, W))
        N = x.numel() // (H * W)
        out = torch.empty(*x.shape[:-2], H - 10, W - 10, device=x.device, dtype=torch.float32)
        _lib.check(_call(("gauss11_filter", (N, H, W)), lib.kmu_gauss11_filter, _ptr(x), _ptr(taps), _ptr(out), N, H, W, 0, _stream()),
                   "kmu_gauss11_filter")
        ctx.save_for_backward(taps)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (taps,) = ctx.saved_tensors
        g = _f32c(g, "grad")
        H, W = g.shape[-2:]
        N = g.numel() // (H * W)
        dx = torch.empty(*g.shape[:-2], H + 10, W + 10, device=g.device, dtype=torch.float32)
        _lib.check(_call(("gauss11_adjoint", (N, H, W)), lib.kmu_gauss11_filter, _ptr(g), _ptr(taps), _ptr(dx), N, H, W, 1, _stream()),
                   "kmu_gauss11_filter")
        return dx, None


def gauss11(x, taps):
    return Gauss11Fn.apply(x, taps)


class HybridLossFn(torch.autograd.Function):
    """train_shanghai.py:298-325 as 6 launches forward / 3 backward (csrc/hybrid_loss.hip + the gauss11 window)."""

    @staticmethod
    def forward(ctx, pred, target, taps, alpha):
        lib = _lib.load()
        pred, target = _f32c(pred, "pred"), _f32c(target, "target")
        if pred.shape != target.shape or pred.dim() != 4:
            raise RuntimeError("hybrid_loss: pred %s and target %s must be equal-shaped [B,C,H,W]" % (tuple(pred.shape), tuple(target.shape)))
        B, C, H, W = pred.shape
        N, dev, st = B * C, pred.device, _stream()
        part = torch.empty(6 * lib.kmu_hybrid_loss_blocks(N, H, W), device=dev, dtype=torch.float32)
        stats = torch.empty(8, device=dev, dtype=torch.float32)
        _lib.check(_call(("hybrid_loss_stats", (N, H, W)), lib.kmu_hybrid_loss_stats, _ptr(pred), _ptr(target), _ptr(part), _ptr(stats), N, H,
                         W, st), "kmu_hybrid_loss_stats")
        stack = torch.empty(5, N, H + 10, W + 10, device=dev, dtype=torch.float32)
        _lib.check(_call(("hybrid_loss_stack", (N, H, W)), lib.kmu_hybrid_loss_stack, _ptr(pred), _ptr(target), _ptr(stats), _ptr(stack), N, H,
                         W, st), "kmu_hybrid_loss_stack")
        filt = torch.empty(5, N, H, W, device=dev, dtype=torch.float32)
        _lib.check(_call(("gauss11_filter", (5 * N, H + 10, W + 10)), lib.kmu_gauss11_filter, _ptr(stack), _ptr(taps), _ptr(filt), 5 * N,
                         H + 10, W + 10, 0, st), "kmu_gauss11_filter")
        _lib.check(_call(("hybrid_loss_combine", (N, H, W)), lib.kmu_hybrid_loss_combine, _ptr(filt), _ptr(part), _ptr(stats), N, H, W,
                         float(alpha), st), "kmu_hybrid_loss_combine")
        ctx.save_for_backward(pred, target, stats, filt, taps)
        ctx.alpha = float(alpha)
        return stats[7].clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pred, target, stats, filt, taps = ctx.saved_tensors
        B, C, H, W = pred.shape
        N, dev, st = B * C, pred.device, _stream()
        g = _f32c(g, "grad").reshape(1)
        gmaps = torch.empty(3, N, H, W, device=dev, dtype=torch.float32)
        _lib.check(_call(("hybrid_loss_grad_maps", (N, H, W)), lib.kmu_hybrid_loss_grad_maps, _ptr(filt), _ptr(g), _ptr(gmaps), N, H, W,
                         ctx.alpha, st), "kmu_hybrid_loss_grad_maps")
        q = torch.empty(3, N, H + 10, W + 10, device=dev, dtype=torch.float32)
        _lib.check(_call(("gauss11_adjoint", (3 * N, H, W)), lib.kmu_gauss11_filter, _ptr(gmaps), _ptr(taps), _ptr(q), 3 * N, H, W, 1, st),
                   "kmu_gauss11_filter")
        dpred = torch.empty_like(pred)
        _lib.check(_call(("hybrid_loss_grad_input", (N, H, W)), lib.kmu_hybrid_loss_grad_input, _ptr(pred), _ptr(target), _ptr(stats), _ptr(q),
                         _ptr(g), _ptr(dpred), N, H, W, ctx.alpha, st), "kmu_hybrid_loss_grad_input")
        return dpred, None, None, None


def hybrid_loss(pred, target, taps, alpha=0.7):
    return HybridLossFn.apply(pred, target, taps, alpha)


# ------------------------------------------------------------------------------------------ BN + ReLU + blend
class BnBlendFn(torch.autograd.Function):
    """out = x + sigmoid(alpha[row]) * (f(t) - x),  f = relu?(BatchNorm2d(t)) or identity
    (EfficientViMBlock.forward, efficient_vim_init.py:81-97, with ConvLayer2D's norm/act, vim_utils_init.py:62-89)."""

    @staticmethod
    def forward(ctx, t, x, gamma, beta, a_row, running_mean, running_var, momentum, eps, relu, training, nbt=None):
        lib = _lib.load()
        t = _f32c(t, "t")
        x = _f32c(x, "x") if x is not None else None
        B, C = t.shape[:2]
        HW = t.numel() // (B * C)
        dev = t.device
        has_bn, has_blend = gamma is not None, a_row is not None
        if has_blend:
            a_row = _f32c(a_row, "alpha row")
        out = torch.empty_like(t)
        S = lib.kmu_bn_blend_splits(B, HW)
        stats = torch.empty(C, 2, device=dev, dtype=torch.float32) if has_bn else None
        ws = torch.empty(C * S * 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("bn_blend_fwd", (B, C, HW)), lib.kmu_bn_blend_fwd, _ptr(t), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(a_row),
                         _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), int(relu), int(training), _ptr(out),
                         _ptr(stats), _ptr(ws), _ptr(nbt), B, C, HW, _stream()), "kmu_bn_blend_fwd")
        if relu and not has_blend:
            _tap_relu(out)
        ctx.save_for_backward(t, x, gamma, beta, a_row, stats)
        ctx.cfg = (int(relu), int(training), B, C, HW, S)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        t, x, gamma, beta, a_row, stats = ctx.saved_tensors
        relu, training, B, C, HW, S = ctx.cfg
        g = _f32c(g, "grad")
        dev = t.device
        has_bn, has_blend = gamma is not None, a_row is not None
        dt = torch.empty_like(t)
        dx = torch.empty_like(t) if has_blend else None
        dg = torch.empty(C, device=dev, dtype=torch.float32) if has_bn else None
        db = torch.empty(C, device=dev, dtype=torch.float32) if has_bn else None
        da = torch.empty(C, device=dev, dtype=torch.float32) if has_blend else None
        ws = torch.empty(C * S * 3, device=dev, dtype=torch.float32)
        _lib.check(_call(("bn_blend_bwd", (B, C, HW)), lib.kmu_bn_blend_bwd, _ptr(g), _ptr(t), _ptr(x), _ptr(gamma), _ptr(beta),
                         _ptr(a_row), _ptr(stats), relu, training, _ptr(dt), _ptr(dx), _ptr(dg), _ptr(db), _ptr(da), _ptr(ws),
                         B, C, HW, _stream()), "kmu_bn_blend_bwd")
        return dt, dx, dg, db, da, None, None, None, None, None, None, None


def bn_blend(t, x=None, bn=None, alpha=None, row=0, relu=False):
    """t: conv output; x: blend partner (or None); bn: an nn.BatchNorm2d (or None); alpha: the RAW layer-scale --
    either the [4, C] parameter (row selects) or an already selected [C] row (or None for no blend)."""
    a_row = None if alpha is None else (alpha if alpha.dim() == 1 else alpha[row])
    if bn is None:
        return BnBlendFn.apply(t, x, None, None, a_row, None, None, 0.0, 0.0, relu, False)
    training = bn.training
    nbt = bn.num_batches_tracked if training and bn.track_running_stats else None     # incremented by the kernel
    return BnBlendFn.apply(t, x, bn.weight, bn.bias, a_row, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, training, nbt)


# ------------------------------------------------------------------------------------------ EfficientViMBlock stages
# Composite stages  out = x + sigmoid(a) (f(x) - x)  with f = BN(dwconv3x3(x)) or f = BN(fc2(ReLU(BN(fc1(x))))).
# As separate autograd nodes, x receives two full-size gradients (through f and through the blend) that the engine sums
# with one `add` kernel per stage (45 per step for these two stage types); here the last backward kernel of the f branch
# (depthwise backward-data / fc1's input gradient) adds the blend gradient in its epilogue.
def _k_bn_fwd(lib, t, x, gamma, beta, a_row, rm, rv, momentum, eps, relu, training, nbt, tap_groups=1):
    B, C = t.shape[:2]
    HW = t.numel() // (B * C)
    out = torch.empty_like(t)
    S = lib.kmu_bn_blend_splits(B, HW)
    stats = torch.empty(C, 2, device=t.device, dtype=torch.float32)
    ws = torch.empty(C * S * 3, device=t.device, dtype=torch.float32)
    _lib.check(_call(("bn_blend_fwd", (B, C, HW)), lib.kmu_bn_blend_fwd, _ptr(t), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(a_row), _ptr(rm),
                     _ptr(rv), float(momentum), float(eps), int(relu), int(training), _ptr(out), _ptr(stats), _ptr(ws), _ptr(nbt), B, C, HW,
                     _stream()), "kmu_bn_blend_fwd")
    if relu and a_row is None:
        _tap_relu(out, tap_groups)
    return out, stats


def _k_bn_fwd_pre(lib, t, x, gamma, beta, a_row, rm, rv, momentum, eps, relu, nbt, part, S):
    """Train-mode BatchNorm (+ReLU / blend) whose statistics partials were left by the kernel that produced t: one launch."""
    B, C = t.shape[:2]
    HW = t.numel() // (B * C)
    out = torch.empty_like(t)
    stats = torch.empty(C, 2, device=t.device, dtype=torch.float32)
    _lib.check(_call(("bn_blend_fwd_pre", (B, C, HW)), lib.kmu_bn_blend_fwd_pre, _ptr(t), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(a_row),
                     _ptr(rm), _ptr(rv), float(momentum), float(eps), int(relu), _ptr(out), _ptr(stats), _ptr(part), S, _ptr(nbt), B, C, HW,
                     _stream()), "kmu_bn_blend_fwd_pre")
    if relu and a_row is None:
        _tap_relu(out)
    return out, stats


def _k_pw_fwd_bn(lib, x, w, x_blend, gamma, beta, a_row, rm, rv, momentum, eps, relu, training, nbt):
    """1x1 conv -> BatchNorm (-> ReLU / blend): in train mode the conv leaves the statistics partials (csrc/pwconv.hip)."""
    B, ci, H, W = x.shape
    co, P = w.shape[0], H * W
    S = lib.kmu_pwconv_stats_partials(B, P) if training else 0
    if not S:
        z = _k_pw_fwd(lib, x, w)
        out, stats = _k_bn_fwd(lib, z, x_blend, gamma, beta, a_row, rm, rv, momentum, eps, relu, training, nbt)
        return z, out, stats
    z = torch.empty(B, co, H, W, device=x.device, dtype=torch.float32)
    part = torch.empty(co * S * 2, device=x.device, dtype=torch.float32)
    _lib.check(_call(("pwconv_fwd", (B, ci, co, P)), lib.kmu_pwconv_fwd_stats, _ptr(x), _ptr(w), None, _ptr(z), _ptr(part), B, ci, co, P, 0,
                     _stream()), "kmu_pwconv_fwd_stats")
    out, stats = _k_bn_fwd_pre(lib, z, x_blend, gamma, beta, a_row, rm, rv, momentum, eps, relu, nbt, part, S)
    return z, out, stats


def _k_bn_bwd(lib, g, t, x, gamma, beta, a_row, stats, relu, training):
    B, C = t.shape[:2]
    HW = t.numel() // (B * C)
    dev = t.device
    S = lib.kmu_bn_blend_splits(B, HW)
    dt = torch.empty_like(t)
    dx = torch.empty_like(t) if a_row is not None else None
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    da = torch.empty(C, device=dev) if a_row is not None else None
    ws = torch.empty(C * S * 3, device=dev, dtype=torch.float32)
    _lib.check(_call(("bn_blend_bwd", (B, C, HW)), lib.kmu_bn_blend_bwd, _ptr(g), _ptr(t), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(a_row),
                     _ptr(stats), int(relu), int(training), _ptr(dt), _ptr(dx), _ptr(dg), _ptr(db), _ptr(da), _ptr(ws), B, C, HW, _stream()),
               "kmu_bn_blend_bwd")
    return dt, dx, dg, db, da


def _k_pw_fwd(lib, x, w):
    B, ci, H, W = x.shape
    co = w.shape[0]
    y = torch.empty(B, co, H, W, device=x.device, dtype=torch.float32)
    _lib.check(_call(("pwconv_fwd", (B, ci, co, H * W)), lib.kmu_pwconv_fwd, _ptr(x), _ptr(w), None, _ptr(y), B, ci, co, H * W, 0, _stream()),
               "kmu_pwconv_fwd")
    return y


def _k_pw_wgrad(lib, x, gy, dw):
    B, ci, H, W = x.shape
    _pw_wgrad_call(lib, x, gy, dw, None, B, ci, gy.shape[1], H * W, 0)
    return dw


class DwBnBlendFn(torch.autograd.Function):
    """x + sigmoid(a) (BatchNorm2d(dwconv3x3(x)) - x): EfficientViMBlock's dwconv1 / dwconv2 stages
    (efficient_vim_init.py:85,93) as one autograd node."""

    @staticmethod
    def forward(ctx, x, w_dw, gamma, beta, a_row, rm, rv, momentum, eps, training, nbt):
        ctx.defer_wgrad = _leaf(w_dw)
        lib = _lib.load()
        x, w, a_row = _f32c(x, "x"), _f32c(w_dw, "weight"), _f32c(a_row, "alpha row")
        B, C, H, W = x.shape
        t = torch.empty_like(x)
        S = lib.kmu_dwconv3x3_stats_partials(B, C, H, W) if training else 0
        if S:       # the stencil leaves the BatchNorm statistics partials: no separate statistics pass
            part = torch.empty(C * S * 2, device=x.device, dtype=torch.float32)
            _lib.check(_call(("dwconv3x3_fwd", (B, C, H, W)), lib.kmu_dwconv3x3_fwd_stats, _ptr(x), _ptr(w), None, _ptr(t), _ptr(part), B, C,
                             H, W, _stream()), "kmu_dwconv3x3_fwd_stats")
            out, stats = _k_bn_fwd_pre(lib, t, x, gamma, beta, a_row, rm, rv, momentum, eps, 0, nbt, part, S)
        else:
            _lib.check(_call(("dwconv3x3_fwd", (B, C, H, W)), lib.kmu_dwconv3x3_fwd, _ptr(x), _ptr(w), None, _ptr(t), B, C, H, W, _stream()),
                       "kmu_dwconv3x3_fwd")
            out, stats = _k_bn_fwd(lib, t, x, gamma, beta, a_row, rm, rv, momentum, eps, 0, training, nbt)
        ctx.save_for_backward(x, w, t, gamma, beta, a_row, stats)
        ctx.cfg = (int(training), tuple(w_dw.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w, t, gamma, beta, a_row, stats = ctx.saved_tensors
        training, wshape = ctx.cfg
        B, C, H, W = x.shape
        st = _stream()
        if DWBN_FUSED and W % 4 == 0 and C <= 4096:
            # BatchNorm's backward folded into the transposed stencil: reduce (partials), then dx = dwconv^T(dt) + (1 - a) g with dt
            # formed on the fly from (g, t) -- no dt / dxb tensors, one launch less on the activation-gradient chain
            g = _f32c(g, "grad")
            dev = x.device
            S = lib.kmu_bn_blend_splits(B, H * W)
            part = torch.empty(C * S * 3, device=dev, dtype=torch.float32)
            _lib.check(_call(("bn_blend_bwd_partials", (B, C, H * W)), lib.kmu_bn_blend_bwd_partials, _ptr(g), _ptr(t), _ptr(x), _ptr(gamma),
                             _ptr(beta), _ptr(a_row), _ptr(stats), 0, _ptr(part), B, C, H * W, st), "kmu_bn_blend_bwd_partials")
            dx = torch.empty_like(x)
            dg, db, da = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(3))
            if DWBN_ALL and B <= 65535:
                # ... and the weight-gradient taps from the same pass: the deferred job is only the column sum of the partials
                dwp = torch.empty(lib.kmu_dwconv3x3_partials(B), C, 9, device=dev, dtype=torch.float32)
                _lib.check(_call(("dwconv3x3_bn_bwd_all", (B, C, H, W)), lib.kmu_dwconv3x3_bn_bwd_all, _ptr(g), _ptr(t), _ptr(x), _ptr(w),
                                 _ptr(gamma), _ptr(a_row), _ptr(stats), _ptr(part), S, training, _ptr(dx), _ptr(dg), _ptr(db), _ptr(da),
                                 _ptr(dwp), B, C, H, W, st), "kmu_dwconv3x3_bn_bwd_all")
                dw = torch.empty(C, 9, device=dev, dtype=torch.float32)
                _wgrad(lambda: colsum(dwp, outs=[dw]), ctx.defer_wgrad)
                return dx, dw.view(wshape), dg, db, da, None, None, None, None, None, None
            cst = torch.empty(C, 4, device=dev, dtype=torch.float32)
            _lib.check(_call(("dwconv3x3_bn_bwd_data", (B, C, H, W)), lib.kmu_dwconv3x3_bn_bwd_data, _ptr(g), _ptr(t), _ptr(w), _ptr(gamma),
                             _ptr(a_row), _ptr(stats), _ptr(part), S, training, _ptr(dx), _ptr(dg), _ptr(db), _ptr(da), _ptr(cst), B, C, H, W,
                             st), "kmu_dwconv3x3_bn_bwd_data")
            dw = torch.empty(C, 9, device=dev, dtype=torch.float32)

            def job():
                P = lib.kmu_dwconv3x3_partials(B)
                dwp = torch.empty(P, C, 9, device=dev, dtype=torch.float32)
                _lib.check(_call(("dwconv3x3_bn_bwd_weight", (B, C, H, W)), lib.kmu_dwconv3x3_bn_bwd_weight, _ptr(x), _ptr(g), _ptr(t), _ptr(cst),
                                 _ptr(dwp), B, C, H, W, _stream()), "kmu_dwconv3x3_bn_bwd_weight")
                colsum(dwp, outs=[dw])
            _wgrad(job, ctx.defer_wgrad)
            return dx, dw.view(wshape), dg, db, da, None, None, None, None, None, None
        dt, dxb, dg, db, da = _k_bn_bwd(lib, _f32c(g, "grad"), t, x, gamma, beta, a_row, stats, 0, training)
        dx = torch.empty_like(x)
        _lib.check(_call(("dwconv3x3_bwd_data", (B, C, H, W)), lib.kmu_dwconv3x3_bwd_data_add, _ptr(dt), _ptr(w), _ptr(dxb), _ptr(dx), B, C, H, W,
                         st), "kmu_dwconv3x3_bwd_data_add")
        dw = torch.empty(C, 9, device=x.device, dtype=torch.float32)

        def job():
            P = lib.kmu_dwconv3x3_partials(B)
            dwp = torch.empty(P, C, 9, device=x.device, dtype=torch.float32)
            _lib.check(_call(("dwconv3x3_bwd_weight", (B, C, H, W)), lib.kmu_dwconv3x3_bwd_weight, _ptr(x), _ptr(dt), _ptr(dwp), None, B, C, H, W,
                             _stream()), "kmu_dwconv3x3_bwd_weight")
            colsum(dwp, outs=[dw])
        _wgrad(job, ctx.defer_wgrad)
        return dx, dw.view(wshape), dg, db, da, None, None, None, None, None, None


class FfnBlendFn(torch.autograd.Function):
    """x + sigmoid(a) (BN2(fc2(ReLU(BN1(fc1(x))))) - x): EfficientViMBlock's FFN stage (efficient_vim_init.py:96;
    FFN = two bias-free 1x1 ConvLayer2D, vim_utils_init.py:62-89,122-130) as one autograd node."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, mom1, eps1, nbt1, w2, g2, b2, rm2, rv2, mom2, eps2, nbt2, a_row, training):
        ctx.defer_wgrad = _leaf(w1, w2)
        lib = _lib.load()
        x, a_row = _f32c(x, "x"), _f32c(a_row, "alpha row")
        hid, C = w1.shape[0], w1.shape[1]
        w1c, w2c = _f32c(w1, "fc1 weight").view(hid, C), _f32c(w2, "fc2 weight").view(C, hid)
        z1, h, st1 = _k_pw_fwd_bn(lib, x, w1c, None, g1, b1, None, rm1, rv1, mom1, eps1, 1, training, nbt1)
        z2, out, st2 = _k_pw_fwd_bn(lib, h, w2c, x, g2, b2, a_row, rm2, rv2, mom2, eps2, 0, training, nbt2)
        ctx.save_for_backward(x, w1c, z1, st1, h, w2c, z2, st2, g1, b1, g2, b2, a_row)
        ctx.cfg = (int(training), tuple(w1.shape), tuple(w2.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w1, z1, st1, h, w2, z2, st2, g1, b1, g2, b2, a_row = ctx.saved_tensors
        training, s1, s2 = ctx.cfg
        B, C, H, W = x.shape
        hid, P, st = w1.shape[0], H * W, _stream()
        dz2, dxb, dg2, db2, da = _k_bn_bwd(lib, _f32c(g, "grad"), z2, x, g2, b2, a_row, st2, 0, training)
        dh = torch.empty_like(h)
        _lib.check(_call(("pwconv_bwd_input", (B, hid, C, P)), lib.kmu_pwconv_bwd_input, _ptr(dz2), _ptr(w2), None, _ptr(dh), B, hid, C, P, 0, st),
                   "kmu_pwconv_bwd_input")
        dw2 = torch.empty(C, hid, device=x.device, dtype=torch.float32)
        _wgrad(lambda: _k_pw_wgrad(lib, h, dz2, dw2), ctx.defer_wgrad)
        dz1, _, dg1, db1, _ = _k_bn_bwd(lib, dh, z1, None, g1, b1, None, st1, 1, training)
        dx = torch.empty_like(x)
        _lib.check(_call(("pwconv_bwd_input", (B, C, hid, P)), lib.kmu_pwconv_bwd_input_add, _ptr(dz1), _ptr(w1), _ptr(dxb), _ptr(dx), B, C, hid, P,
                         st), "kmu_pwconv_bwd_input_add")
        dw1 = torch.empty(hid, C, device=x.device, dtype=torch.float32)
        _wgrad(lambda: _k_pw_wgrad(lib, x, dz1, dw1), ctx.defer_wgrad)
        return (dx, dw1.view(s1), dg1, db1, None, None, None, None, None, dw2.view(s2), dg2, db2, None, None, None, None, None, da, None)


# The FFN stage as recompute kernels (csrc/ffn_fused.hip): nothing 4C wide is stored.  KMU_FFN_FUSED=0 keeps the pointwise-conv +
# BatchNorm kernels (FfnBlendFn), which also serve the shapes the fused kernels do not cover (C not in {16, 32, 64}, H*W % 64 != 0).
FFN_FUSED = True
TAIL_FUSED = True     # EnhancedViMBlock's tail FFN as one recompute launch each way
DWBN_FUSED = True     # dwconv stage backward: BatchNorm folded into the transposed stencil
DWBN_ALL = True       # ... and the weight-gradient taps taken in the same launch
_FFN_STAGES_F = ("ffn_fwd_stats", "ffn_fwd_main", "ffn_fwd_apply")
_FFN_STAGES_B = ("ffn_bwd_red", "ffn_bwd_mid", "ffn_bwd_in")


class FfnFusedFn(torch.autograd.Function):
    """x + sigmoid(a) (BN2(fc2(ReLU(BN1(fc1(x))))) - x), EfficientViMBlock's FFN stage (efficient_vim_init.py:96; FFN = two bias-free
    1x1 ConvLayer2D, vim_utils_init.py:62-89,122-130), as 3 + 3 recompute launches: saved for backward are x, z2 = fc2(...) and the two
    BatchNorms' (mean, rstd) -- no 4C-wide tensor, forward or backward; both weight gradients leave per-workgroup slabs whose column
    sums run off the activation-gradient chain."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, rm1, rv1, mom1, eps1, nbt1, w2, g2, b2, rm2, rv2, mom2, eps2, nbt2, a_row, training):
        ctx.defer_wgrad = _leaf(w1, w2)
        lib = _lib.load()
        x, a_row = _f32c(x, "x"), _f32c(a_row, "alpha row")
        B, C, H, W = x.shape
        P, hid, dev = H * W, w1.shape[0], x.device
        w1c, w2c = _f32c(w1, "fc1 weight").view(hid, C), _f32c(w2, "fc2 weight").view(C, hid)
        g1, b1, g2, b2 = _f32c(g1, "bn1 weight"), _f32c(b1, "bn1 bias"), _f32c(g2, "bn2 weight"), _f32c(b2, "bn2 bias")
        z2, out = torch.empty_like(x), torch.empty_like(x)
        st1 = torch.empty(hid, 2, device=dev, dtype=torch.float32)
        st2 = torch.empty(C, 2, device=dev, dtype=torch.float32)
        nbytes = lib.kmu_ffn_fused_fwd_ws_bytes(B, C, P)
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        h = torch.empty(B, hid, H, W, device=dev, dtype=torch.float32) if RELU_TAP is not None else None
        tr = int(bool(training))
        st = _stream()
        for stage, nm in enumerate(_FFN_STAGES_F):
            if stage == 0 and not tr:
                continue
            _lib.check(_call((nm, (B, C, P)), lib.kmu_ffn_fused_fwd, _ptr(x), _ptr(w1c), _ptr(g1), _ptr(b1), _ptr(rm1), _ptr(rv1),
                             _ptr(nbt1 if tr else None), float(mom1), float(eps1), _ptr(w2c), _ptr(g2), _ptr(b2), _ptr(rm2), _ptr(rv2),
                             _ptr(nbt2 if tr else None), float(mom2), float(eps2), _ptr(a_row), tr, _ptr(z2), _ptr(out), _ptr(st1),
                             _ptr(st2), _ptr(h), _ptr(ws), nbytes, B, C, P, stage, st), "kmu_ffn_fused_fwd")
        if h is not None:
            _tap_relu(h)
        ctx.save_for_backward(x, z2, w1c, w2c, g1, b1, g2, b2, st1, st2, a_row)
        ctx.cfg = (tr, tuple(w1.shape), tuple(w2.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, z2, w1, w2, g1, b1, g2, b2, st1, st2, a_row = ctx.saved_tensors
        tr, s1, s2 = ctx.cfg
        g = _f32c(g, "grad")
        B, C, H, W = x.shape
        P, hid, dev = H * W, w1.shape[0], x.device
        rows1, rows2 = lib.kmu_ffn_fused_rows(B, C, P, 1), lib.kmu_ffn_fused_rows(B, C, P, 2)
        mk = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        dx, dg1, db1, dg2, db2, da = torch.empty_like(x), mk(hid), mk(hid), mk(C), mk(C), mk(C)
        slab1, slab2 = mk(rows1, hid * C), mk(rows2, C * hid)
        nbytes = lib.kmu_ffn_fused_bwd_ws_bytes(B, C, P)
        ws = mk(nbytes // 4)
        st = _stream()
        for stage, nm in enumerate(_FFN_STAGES_B):
            _lib.check(_call((nm, (B, C, P)), lib.kmu_ffn_fused_bwd, _ptr(g), _ptr(x), _ptr(z2), _ptr(w1), _ptr(g1), _ptr(b1), _ptr(st1),
                             _ptr(w2), _ptr(g2), _ptr(b2), _ptr(st2), _ptr(a_row), tr, _ptr(dx), _ptr(dg1), _ptr(db1), _ptr(dg2), _ptr(db2),
                             _ptr(da), _ptr(slab1), _ptr(slab2), _ptr(ws), nbytes, B, C, P, stage, st), "kmu_ffn_fused_bwd")
        dw1, dw2 = mk(hid, C), mk(C, hid)
        _wgrad(lambda: colsum(slab1, slab2, outs=[dw1.view(-1), dw2.view(-1)]), ctx.defer_wgrad)
        return (dx, dw1.view(s1), dg1, db1, None, None, None, None, None, dw2.view(s2), dg2, db2, None, None, None, None, None, da, None)


def ffn_fused_supported(C, hid, P):
    return FFN_FUSED and bool(_lib.load().kmu_ffn_fused_supported(C, hid, P))


def _bn_pack(bn):
    training = bn.training
    nbt = bn.num_batches_tracked if training and bn.track_running_stats else None
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, nbt, training


def dw_bn_blend(x, conv, bn, a_row):
    g, b, rm, rv, mom, eps, nbt, training = _bn_pack(bn)
    return DwBnBlendFn.apply(x, conv.weight, g, b, a_row, rm, rv, mom, eps, training, nbt)


def ffn_blend(x, fc1, fc2, a_row):
    g1, b1, rm1, rv1, mom1, eps1, nbt1, training = _bn_pack(fc1.norm)
    g2, b2, rm2, rv2, mom2, eps2, nbt2, _ = _bn_pack(fc2.norm)
    if x.is_cuda and ffn_fused_supported(x.shape[1], fc1.conv.out_channels, x.shape[2] * x.shape[3]) and fc2.conv.out_channels == x.shape[1]:
        return FfnFusedFn.apply(x, fc1.conv.weight, g1, b1, rm1, rv1, mom1, eps1, nbt1, fc2.conv.weight, g2, b2, rm2, rv2, mom2, eps2, nbt2,
                                a_row, training)
    return FfnBlendFn.apply(x, fc1.conv.weight, g1, b1, rm1, rv1, mom1, eps1, nbt1, fc2.conv.weight, g2, b2, rm2, rv2, mom2, eps2, nbt2, a_row,
                            training)


# ------------------------------------------------------------------------------------------ sigmoid(q*k)*v
class QkvGateFn(torch.autograd.Function):
    """attn = sigmoid(q*k)*v on the packed qkv conv output (DirectionAttention.forward, KM_UNetV3_SH.py:258-261)."""

    @staticmethod
    def forward(ctx, qkv):
        lib = _lib.load()
        qkv = _f32c(qkv, "qkv")
        B, C3, H, W = qkv.shape
        C = C3 // 3
        out = torch.empty(B, C, H, W, device=qkv.device, dtype=torch.float32)
        _lib.check(_call(("qkv_gate_fwd", (B, C, H * W)), lib.kmu_qkv_gate_fwd, _ptr(qkv), _ptr(out), B, C, H * W, _stream()),
                   "kmu_qkv_gate_fwd")
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (qkv,) = ctx.saved_tensors
        g = _f32c(g, "grad")
        B, C3, H, W = qkv.shape
        d = torch.empty_like(qkv)
        _lib.check(_call(("qkv_gate_bwd", (B, C3 // 3, H * W)), lib.kmu_qkv_gate_bwd, _ptr(qkv), _ptr(g), _ptr(d), B, C3 // 3, H * W,
                         _stream()), "kmu_qkv_gate_bwd")
        return d


def qkv_gate(qkv):
    return QkvGateFn.apply(qkv)


# ------------------------------------------------------------------------------------------ pointwise conv
def pwconv_supported(ci, co, hw):
    return ci % 16 == 0 and co % 16 == 0 and 0 < ci <= 1024 and 0 < co <= 1024 and hw % 64 == 0


class PwConvFn(torch.autograd.Function):
    """y = conv1x1(act(x), weight) + bias with act = identity | exact GELU (KM_UNetV3_SH.py:120-124,174,221;
    vim_utils_init.py:62-89); bias, GELU, GELU' and the bias gradient are folded into the three kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, act_in):
        ctx.defer_wgrad = _leaf(weight, bias)
        lib = _lib.load()
        x = _f32c(x, "x")
        co = weight.shape[0]
        ci = weight.numel() // co              # [Co, Ci, 1, 1], [Co, Ci] or a 3-tap kernel [Co, Ci/3, 3, 1] over stacked taps
        if x.shape[1] != ci:
            raise RuntimeError("pwconv: weight %s does not contract the %d input channels" % (tuple(weight.shape), x.shape[1]))
        w = _f32c(weight, "weight").view(co, ci)
        B, _, H, W = x.shape
        y = torch.empty(B, co, H, W, device=x.device, dtype=torch.float32)
        _lib.check(_call(("pwconv_fwd", (B, ci, co, H * W)), lib.kmu_pwconv_fwd, _ptr(x), _ptr(w),
                         _ptr(None if bias is None else _f32c(bias, "bias")), _ptr(y), B, ci, co, H * W, int(act_in), _stream()),
                   "kmu_pwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (bias is not None, int(act_in), tuple(weight.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        has_bias, act_in, wshape = ctx.cfg
        g = _f32c(g, "grad")
        B, ci, H, W = x.shape
        co, P = w.shape[0], H * W
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(_call(("pwconv_bwd_input", (B, ci, co, P)), lib.kmu_pwconv_bwd_input, _ptr(g), _ptr(w),
                             _ptr(x) if act_in else None, _ptr(dx), B, ci, co, P, act_in, _stream()), "kmu_pwconv_bwd_input")
        dw = torch.empty(co, ci, device=x.device, dtype=torch.float32)
        db = torch.empty(co, device=x.device, dtype=torch.float32) if has_bias else None

        _wgrad(lambda: _pw_wgrad_call(lib, x, g, dw, db, B, ci, co, P, act_in), ctx.defer_wgrad)
        return dx, dw.view(wshape), db, None


def pwconv(x, weight, bias=None, act_in=False):
    return PwConvFn.apply(x, weight, bias, act_in)


class MeanPwConvFn(torch.autograd.Function):
    """(mean_hw(x) [B,C], conv1x1(x, weight) + bias) as ONE autograd node: DirectionAttention.forward (KM_UNetV3_SH.py:231, :258)
    pools x for its channel gate and projects the same x to q, k, v.  As two nodes the backward ran a scale launch on d mean, the
    1x1 input gradient, and an ATen broadcast add of the two at the fan-in; here d mean / HW rides in the input-gradient kernel's
    epilogue as a per-(sample, channel) constant."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.defer_wgrad = _leaf(weight, bias)
        lib = _lib.load()
        x = _f32c(x, "x")
        co, ci = weight.shape[:2]
        w = _f32c(weight, "weight").view(co, ci)
        B, _, H, W = x.shape
        st = _stream()
        pooled = torch.empty(B, ci, device=x.device, dtype=torch.float32)
        _lib.check(_call(("mean_rows", (B, ci, H * W)), lib.kmu_mean_rows, _ptr(x), None, None, _ptr(pooled), B, ci, H * W, 1, st), "kmu_mean_rows")
        y = torch.empty(B, co, H, W, device=x.device, dtype=torch.float32)
        _lib.check(_call(("pwconv_fwd", (B, ci, co, H * W)), lib.kmu_pwconv_fwd, _ptr(x), _ptr(w),
                         _ptr(None if bias is None else _f32c(bias, "bias")), _ptr(y), B, ci, co, H * W, 0, st), "kmu_pwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (bias is not None, tuple(weight.shape))
        ctx.set_materialize_grads(False)
        return pooled, y

    @staticmethod
    def backward(ctx, gp, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        has_bias, wshape = ctx.cfg
        B, ci, H, W = x.shape
        co, P = w.shape[0], H * W
        if g is None:         # only the pooled output was used
            dx = None
            if gp is not None and ctx.needs_input_grad[0]:
                dx = (_f32c(gp, "grad") * (1.0 / P)).view(B, ci, 1, 1).expand(B, ci, H, W)
            return dx, None, None
        g = _f32c(g, "grad")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if gp is None:
                _lib.check(_call(("pwconv_bwd_input", (B, ci, co, P)), lib.kmu_pwconv_bwd_input, _ptr(g), _ptr(w), None, _ptr(dx), B, ci, co,
                                 P, 0, _stream()), "kmu_pwconv_bwd_input")
            else:
                _lib.check(_call(("pwconv_bwd_input", (B, ci, co, P)), lib.kmu_pwconv_bwd_input_rowadd, _ptr(g), _ptr(w),
                                 _ptr(_f32c(gp, "grad")), 1.0 / P, _ptr(dx), B, ci, co, P, _stream()), "kmu_pwconv_bwd_input_rowadd")
        dw = torch.empty(co, ci, device=x.device, dtype=torch.float32)
        db = torch.empty(co, device=x.device, dtype=torch.float32) if has_bias else None
        _wgrad(lambda: _pw_wgrad_call(lib, x, g, dw, db, B, ci, co, P, 0), ctx.defer_wgrad)
        return dx, dw.view(wshape), db


def mean_pwconv(x, weight, bias=None):
    return MeanPwConvFn.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------ wavelet pooling front end
_ZEROS = {}


def _const_zeros(like):
    """A shared, never-written all-zero tensor (gradients that are exactly zero by construction): no fill launch."""
    key = (tuple(like.shape), like.device)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(like.shape, device=like.device, dtype=torch.float32)
    return _ZEROS[key]


class IwpFrontFn(torch.autograd.Function):
    """cat[LL, mean(cat[LH, HL, HH] * Softmax2d(conv1x1(.)))] of WPL/iwp.py:124-130.  hf_weight / hf_bias are the
    parameters of high_freq_conv: a softmax over its single output channel is identically 1, so they do not enter
    the value and their gradient is exactly zero (returned as such so that they stay "live" for AdamW's decay)."""

    @staticmethod
    def forward(ctx, x, hf_weight, hf_bias, ct=None):
        lib = _lib.load()
        x = _f32c(x, "x")
        B, C, H, W = x.shape
        ct = C + 1 if ct is None else int(ct)            # > C+1: zero channels appended (channel padding for pwconv)
        out = torch.empty(B, ct, H // 2, W // 2, device=x.device, dtype=torch.float32)
        _lib.check(_call(("iwp_front_fwd", (B, C, H, W)), lib.kmu_iwp_front_fwd, _ptr(x), _ptr(out), B, C, ct, H, W, _stream()),
                   "kmu_iwp_front_fwd")
        ctx.dims = (B, C, H, W, ct)
        ctx.zeros = (_const_zeros(hf_weight), None if hf_bias is None else _const_zeros(hf_bias))
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, C, H, W, ct = ctx.dims
        g = _f32c(g, "grad")
        dx = torch.empty(B, C, H, W, device=g.device, dtype=torch.float32)
        _lib.check(_call(("iwp_front_bwd", (B, C, H, W)), lib.kmu_iwp_front_bwd, _ptr(g), _ptr(dx), B, C, ct, H, W, _stream()),
                   "kmu_iwp_front_bwd")
        return dx, ctx.zeros[0], ctx.zeros[1], None


def iwp_front(x, hf_weight, hf_bias, ct=None):
    return IwpFrontFn.apply(x, hf_weight, hf_bias, ct)


# ------------------------------------------------------------------------------------------ squeeze-excite gates
_ACT1 = {"gelu": 0, "silu": 1, "relu": 2}
_ACT2 = {"sigmoid": 0, "softmax": 1}


class GateMlpFn(torch.autograd.Function):
    """g = act2(W2 act1(W1 p + b1) + b2) on pooled [B, I] vectors (KM_UNetV3_SH.py:111-117,231-236,320-325,342-347):
    one launch forward, one backward (weight gradients summed over the batch in-kernel)."""

    @staticmethod
    def forward(ctx, p, w1, b1, w2, b2, act1, act2):
        lib = _lib.load()
        p = _f32c(p, "pooled input")
        B, I = p.shape
        H, O = w1.shape[0], w2.shape[0]
        w1c, w2c = _f32c(w1, "w1").view(H, I), _f32c(w2, "w2").view(O, H)
        z1 = torch.empty(B, H, device=p.device, dtype=torch.float32)
        g = torch.empty(B, O, device=p.device, dtype=torch.float32)
        _lib.check(_call(("gate_mlp_fwd", (B, I, H, O)), lib.kmu_gate_mlp_fwd, _ptr(p), _ptr(w1c), _ptr(None if b1 is None else _f32c(b1, "b1")),
                         _ptr(w2c), _ptr(None if b2 is None else _f32c(b2, "b2")), _ptr(z1), _ptr(g), B, I, H, O, _ACT1[act1], _ACT2[act2],
                         _stream()), "kmu_gate_mlp_fwd")
        if act1 == "relu":
            _tap_relu(z1)
        ctx.save_for_backward(p, w1c, w2c, z1, g)
        ctx.cfg = (_ACT1[act1], _ACT2[act2], b1 is not None, b2 is not None, tuple(w1.shape), tuple(w2.shape))
        return g

    @staticmethod
    def backward(ctx, dg):
        lib = _lib.load()
        p, w1, w2, z1, g = ctx.saved_tensors
        a1, a2, hb1, hb2, s1, s2 = ctx.cfg
        dg = _f32c(dg, "grad")
        B, I = p.shape
        H, O = w1.shape[0], w2.shape[0]
        dev = p.device
        dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        dw1, dw2 = torch.empty(H, I, device=dev), torch.empty(O, H, device=dev)
        db1 = torch.empty(H, device=dev) if hb1 else None
        db2 = torch.empty(O, device=dev) if hb2 else None
        _lib.check(_call(("gate_mlp_bwd", (B, I, H, O)), lib.kmu_gate_mlp_bwd, _ptr(p), _ptr(w1), _ptr(w2), _ptr(z1), _ptr(g), _ptr(dg),
                         _ptr(dp), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), B, I, H, O, a1, a2, _stream()), "kmu_gate_mlp_bwd")
        return dp, dw1.view(s1), db1, dw2.view(s2), db2, None, None


def gate_mlp(p, w1, b1, w2, b2, act1="gelu", act2="sigmoid"):
    return GateMlpFn.apply(p, w1, b1, w2, b2, act1, act2)


# ------------------------------------------------------------------------------------------ GroupNorm
class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(G, C) forward/backward (KM_UNetV3_SH.py:57,271-273,294,448)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, act=0):
        lib = _lib.load()
        ctx.defer_wgrad = _leaf(gamma, beta)
        x, gamma, beta = _f32c(x, "x"), _f32c(gamma, "weight"), _f32c(beta, "bias")
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        S = lib.kmu_group_norm_splits(HW)
        y = torch.empty_like(x)
        stats = torch.empty(B, G, 2, device=x.device, dtype=torch.float32)
        ws = torch.empty(B * C * S * 2, device=x.device, dtype=torch.float32)
        _lib.check(_call(("group_norm_fwd", (B, C, G, HW)), lib.kmu_group_norm_act_fwd, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y),
                         _ptr(stats), _ptr(ws), B, C, G, HW, float(eps), int(act), _stream()), "kmu_group_norm_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (B, C, G, HW, S, int(act))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, gamma, beta, stats = ctx.saved_tensors
        B, C, G, HW, S, act = ctx.cfg
        g = _f32c(g, "grad")
        dx = torch.empty_like(x)
        dgp = torch.empty(B, C, device=x.device, dtype=torch.float32)
        dbp = torch.empty(B, C, device=x.device, dtype=torch.float32)
        ws = torch.empty(B * C * S * 2, device=x.device, dtype=torch.float32)
        _lib.check(_call(("group_norm_bwd", (B, C, G, HW)), lib.kmu_group_norm_act_bwd, _ptr(x), _ptr(g), _ptr(gamma), _ptr(beta), _ptr(stats),
                         _ptr(dx), _ptr(dgp), _ptr(dbp), _ptr(ws), B, C, G, HW, act, _stream()), "kmu_group_norm_bwd")
        dg, db = torch.empty(C, device=x.device, dtype=torch.float32), torch.empty(C, device=x.device, dtype=torch.float32)
        _wgrad(lambda: colsum(dgp, dbp, outs=[dg, db]), ctx.defer_wgrad)
        return dx, dg, db, None, None, None


def group_norm(x, gn, silu=False, sigmoid=False):
    """gn: an nn.GroupNorm module (parameters + num_groups + eps); silu / sigmoid: the activation folded into the normalisation kernels'
    epilogue (and its derivative into the backward kernels)."""
    return GroupNormFn.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, 1 if silu else (2 if sigmoid else 0))
