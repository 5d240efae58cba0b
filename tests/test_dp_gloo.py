"""CPU, world_size 2, gloo: the data-parallel exchange step (km-unet_amd/dp.py) -- one flat gradient
bucket, parameters that never receive a gradient left out of it, result equal to a single process
on the concatenated batch.  The hot blocks need the GPU, so a small stand-in network with the same
structural quirks (an unused twin branch, a buffer) exercises the host logic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.used = nn.Sequential(nn.Conv2d(5, 8, 3, padding=1), nn.GELU(), nn.Conv2d(8, 5, 3, padding=1))
        self.unused_twin = nn.Conv2d(5, 8, 3, padding=1)       # like StableHybridKANConv.branches.plain
        self.register_buffer("init_pos", torch.ones(3))

    def forward(self, x):
        return torch.sigmoid(self.used(x))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import km_unet_amd
        from km_unet_amd.train import TrainStep, split_frames
        torch.manual_seed(100 + rank)           # deliberately different init per rank: broadcast must fix it
        net = Net()
        data = torch.rand(4, 10, 1, 8, 8, generator=torch.Generator().manual_seed(5))
        shard = data[rank * 2:(rank + 1) * 2]
        step = TrainStep(net, shard, lr=1e-2, loss="mse")    # MSE is batch-separable (HybridLoss min-max-normalises per shard)
        assert step.dp.bucket.numel() == sum(p.numel() for p in net.used.parameters())
        assert all(p.grad is None for p in net.unused_twin.parameters())
        w0 = [p.detach().clone() for p in net.parameters()]
        loss = step(shard)
        flat = step.dp.bucket.flat.clone()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1])            # replicas hold identical averaged grads
        if rank == 0:
            ret["grads"] = flat
            ret["w0"] = w0
            ret["w1"] = [p.detach().clone() for p in net.parameters()]
            ret["loss"] = loss.item()
    finally:
        dist.destroy_process_group()


def test_dp_world2_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    # single-process reference on the full batch with rank 0's initial weights
    net = Net()
    with torch.no_grad():
        for p, w in zip(net.parameters(), ret["w0"]):
            p.copy_(w)
    data = torch.rand(4, 10, 1, 8, 8, generator=torch.Generator().manual_seed(5))
    d = data.squeeze(2)
    loss = torch.nn.functional.mse_loss(net(d[:, :5]), d[:, 5:])
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.used.parameters()])
    assert torch.allclose(ret["grads"], ref, rtol=1e-5, atol=1e-7)
    opt = torch.optim.AdamW(list(net.used.parameters()), lr=1e-2, weight_decay=0.05)
    opt.step()
    for p, w in zip(net.parameters(), ret["w1"]):
        assert torch.allclose(p, w, rtol=1e-5, atol=1e-6)


def _worker_model(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import km_unet_amd
        from km_unet_amd.train import TrainStep
        from oracle.model import KM_UNetV3 as Oracle, fill_parameters
        torch.set_num_threads(2)
        net = fill_parameters(Oracle(num_classes=5), 20 + rank).eval()     # different weights per rank: rank 0's must win
        data = torch.rand(2, 10, 1, 16, 16, generator=torch.Generator().manual_seed(9))
        shard = data[rank:rank + 1]
        step = TrainStep(net, shard, lr=1e-3, loss="mse")
        live, dead = len(step.dp.bucket.params), sum(1 for p in net.parameters() if p.grad is None)
        step(shard)
        flat = step.dp.bucket.flat.clone()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1])
        if rank == 0:
            ret["live"], ret["dead"], ret["numel"], ret["grads"] = live, dead, step.dp.bucket.numel(), flat
    finally:
        dist.destroy_process_group()


def test_dp_world2_real_parameter_structure():
    """The same exchange step on the REAL parameter structure (KM_UNetV3's 714 parameter tensors, 50 of them dead:
    branches.plain, attn, dt_proj -- KM_UNetV3_SH.py:27-34,50-54,163) instead of a stand-in.  The HIP model cannot run in
    this container, so the CPU oracle (same module tree, same state_dict keys) carries the structure; eval mode so that
    BatchNorm uses running statistics and two 1-sample shards equal one 2-sample batch exactly."""
    from oracle.model import KM_UNetV3 as Oracle, fill_parameters
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_model, args=(world, port, ret), nprocs=world, join=True)
    assert ret["live"] == 664 and ret["dead"] == 50
    net = fill_parameters(Oracle(num_classes=5), 20).eval()
    data = torch.rand(2, 10, 1, 16, 16, generator=torch.Generator().manual_seed(9)).squeeze(2)
    torch.nn.functional.mse_loss(net(data[:, :5]), data[:, 5:]).backward()
    from km_unet_amd.dp import branch_adjacent_order     # the bucket keeps the three direction branches' tensors adjacent
    ref = torch.cat([p.grad.reshape(-1) for p in branch_adjacent_order(net, [p for p in net.parameters() if p.grad is not None])])
    assert ref.numel() == ret["numel"]
    err = (ret["grads"] - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err


def test_split_frames_contract():
    """train_shanghai.py:165-167: [B,T,1,H,W] -> input 5 frames, target the rest."""
    import km_unet_amd
    from km_unet_amd.train import split_frames
    data = torch.arange(2 * 10 * 4 * 4, dtype=torch.float32).view(2, 10, 1, 4, 4)
    i, t = split_frames(data)
    assert i.shape == (2, 5, 4, 4) and t.shape == (2, 5, 4, 4)
    assert torch.equal(i[1, 4], data[1, 4, 0]) and torch.equal(t[0, 0], data[0, 5, 0])
