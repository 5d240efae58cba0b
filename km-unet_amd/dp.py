"""Batch-sharded data parallelism: one process per GPU, one flat fp32 gradient bucket, one
sum-all-reduce per step (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The reference has no distributed code at all; this is the only exchange step the
path has: replicas hold the full 6.9 MB of weights, rank r trains on its own batch shard, and
gradients are averaged.  Design notes (MI355X):
  * the gradients of all parameters that can ever receive one live in ONE contiguous buffer
    (`param.grad` are views into it), so the collective is a single ~5 MB message -- latency-bound on
    xGMI, where one large ring/tree call beats many small ones; backward writes into it with one
    multi-tensor copy (DataParallel.backward) instead of 664 per-parameter accumulations;
  * 50 parameter tensors (26 % of the weights: branches.plain, attn, dt_proj -- KM_UNetV3_SH.py:27-34,
    50-54,163) never get a gradient; they are discovered once with a dry backward and left out of the
    bucket (their .grad stays None, AdamW skips them);
  * BatchNorm uses per-replica batch statistics, exactly like the reference on one GPU (no SyncBN,
    DAGEM_md.py:8-12 ignores its sync_bn flag); buffers are broadcast from rank 0 at start-up.
"""
import os

import torch
import torch.distributed as dist


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = self.views[-1]
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def flatten_parameters(self):
        """Move the parameters themselves into one contiguous buffer (`p.data` become views of it, values unchanged)
        and return it as a leaf whose .grad is the gradient bucket: the optimizer then updates ONE tensor -- a single
        fused AdamW launch instead of ~30 multi-tensor launches over 664 small tensors.  Uniform hyper-parameters
        (train_shanghai.py:342 passes model.parameters() as one group), so the update is element-for-element the same."""
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.params]) if self.params else self.flat.clone()
            off = 0
            for p in self.params:
                p.data = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        flat.requires_grad_(True)
        flat.grad = self.flat
        return flat

    def store(self, grads):
        """Write one gradient per parameter into the bucket (a handful of multi-tensor copy launches)."""
        # the multi-tensor fast path needs matching strides for EVERY pair (one permuted gradient sends all 664 copies
        # down the per-tensor path: +2.5 ms per step); stragglers are copied on their own
        fast_v, fast_g = [], []
        for v, g in zip(self.views, grads):
            if g.stride() != v.stride() and g.is_contiguous():
                g = g.view(-1).view(v.shape)        # same memory, canonical strides (size-1 dims may carry any stride): fast path
            if g.stride() == v.stride() and g.dtype == v.dtype:
                fast_v.append(v)
                fast_g.append(g)
            else:
                v.copy_(g)
        if fast_v:
            if fast_v[0].is_cuda and all(g.is_contiguous() and v.is_contiguous() and g.dtype == torch.float32 for v, g in zip(fast_v, fast_g)):
                from . import ops
                ops.copy_multi(fast_v, fast_g)        # 5 launches of a pointer-table copy kernel instead of ATen's 11
            else:
                torch._foreach_copy_(fast_v, fast_g)

    def numel(self):
        return self.flat.numel()


def live_parameters(model, example_input, loss_fn=None):
    """Parameters that receive a gradient from one dry forward/backward (static for this model)."""
    was_training = model.training
    out = model(example_input)
    (loss_fn(out) if loss_fn else out.float().mean()).backward()
    live = [p for p in model.parameters() if p.grad is not None]
    for p in model.parameters():
        p.grad = None
    model.train(was_training)
    return live


def branch_adjacent_order(model, params):
    """The same parameters, reordered so that the corresponding tensors of EnhancedViMBlock's height / width / channel branches
    sit next to each other: FlatGradBucket.flatten_parameters then lays them back to back in the flat buffer, and the stacked
    branch pass (grouped.py) views each triple as one [3, ...] tensor without copying."""
    name_of = {id(p): n for n, p in model.named_parameters()}
    tags = ("height_block.", "width_block.", "channel_block.")
    first, keyed = {}, []
    for idx, p in enumerate(params):
        name, br = name_of.get(id(p), "#%d" % idx), 0
        key = name
        for bi, t in enumerate(tags):
            if t in name:
                key, br = name.replace(t, "@."), bi
                break
        first.setdefault(key, idx)
        keyed.append((first[key], br, idx, p))
    keyed.sort(key=lambda t: t[:3])
    return [t[3] for t in keyed]


class DataParallel:
    """Minimal DDP: broadcast parameters/buffers from rank 0, average gradients after backward."""

    def __init__(self, model, live_params, process_group=None, force_collective=False):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collective: run the all-reduce (and the two-graph step split around it) even with a single rank -- lets the
        # RCCL code path execute on a one-GPU box (RCCL accepts a 1-rank communicator)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        if self.world > 1:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
        self.bucket = FlatGradBucket(branch_adjacent_order(model, live_params))

    def zero_grad(self):
        self.bucket.zero_()

    def backward(self, loss):
        """d loss / d (live parameters) straight into the flat bucket.  `loss.backward()` would run one AccumulateGrad
        node per parameter, each an `add` kernel into its (zeroed) bucket view: 664 launches of ~4 us per step on
        MI355X, 9 % of the step.  autograd.grad hands the gradients back instead and the bucket takes them with
        one multi-tensor copy; nothing needs zeroing because every live parameter is overwritten."""
        if not loss.is_cuda or os.environ.get("KMU_WGRAD_OVERLAP", "1") != "1":
            self.bucket.store(torch.autograd.grad(loss, self.bucket.params))
            return
        # parameter-gradient kernels are queued and dealt onto side streams in batches, off the activation-gradient chain
        # (ops._wgrad); they are joined here, before the bucket copy reads their results
        from . import ops
        ops.WGRAD_OVERLAP = True
        try:
            grads = torch.autograd.grad(loss, self.bucket.params)
        finally:
            ops.WGRAD_OVERLAP = False
            ops.flush_wgrad_jobs(final=True)
        self.bucket.store(grads)

    def all_reduce_grads(self):
        """Average the flat gradient bucket over the ranks: ONE collective.  On RCCL the 1 / world factor rides in the reduction
        itself (ncclAvg) -- no separate division launch between the two captured graphs; gloo (CPU tests) has no averaging
        reduction, so it sums and divides."""
        if not self.collective:
            return
        if dist.get_backend(self.pg) == "nccl":
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.pg)
            if self.world > 1:
                self.bucket.flat.div_(self.world)
