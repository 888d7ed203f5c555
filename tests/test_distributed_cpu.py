"""CPU, world_size = 2 over gloo: the data-parallel gradient exchange (ytvln/distributed.py) is correct by construction.

The bucket reducer and the DataParallel wrapper are device-agnostic; here they are driven by a small CPU network whose
parameters / gradients are views into flat arenas exactly as ytvln.optimization.AdamW lays them out on the GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b, self.c = nn.Linear(6, 8), nn.Linear(8, 8), nn.Linear(8, 3)
        self.unused = nn.Linear(4, 4)          # never touched: must stay out of the arena

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


class ArenaSGD:
    """Stand-in for ytvln.optimization.AdamW's arena protocol on CPU: flat grads, grad_sync hook, grad_scale."""

    def __init__(self, params, lr):
        self.params, self.lr, self.grad_sync, self.grad_scale, self.flat = list(params), lr, None, 1.0, None

    def step(self):
        members = [p for p in self.params if p.grad is not None]
        if self.flat is None:
            off, self.index = 0, {}
            for p in members:
                self.index[id(p)] = (off, p.numel())
                off += (p.numel() + 3) // 4 * 4
            self.flat = torch.zeros(off)
            for p in members:
                o, n = self.index[id(p)]
                self.flat[o:o + n].copy_(p.grad.reshape(-1))
                p.grad = self.flat[o:o + n].view(p.shape)
        if self.grad_sync is not None:
            self.grad_sync(self.flat, [(p,) + self.index[id(p)] for p in members])
        with torch.no_grad():
            for p in members:
                p.add_(p.grad, alpha=-self.lr * self.grad_scale)

    def zero_grad(self):
        if self.flat is not None:
            self.flat.zero_()


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ytvln import distributed as D
    assert D.init_distributed(backend="gloo") == (rank, world)
    torch.manual_seed(100 + rank)                      # different initial weights per rank: wrap must broadcast rank 0's
    net = Net()
    dp = D.DataParallel(net, bucket_bytes=256)         # tiny buckets -> several collectives, exercised in overlap mode
    w0 = [p.detach().clone() for p in net.parameters()]
    gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
    dist.all_gather(gathered, w0[0])
    assert torch.equal(gathered[0], gathered[1]), "parameters must be identical after wrapping"
    opt = ArenaSGD(net.parameters(), lr=0.1)
    dp.attach(opt)
    assert opt.grad_scale == 0.5
    g = torch.Generator().manual_seed(7)
    X = torch.randn(2, 3, 5, 6, generator=g)           # [step, rank, batch, features]
    losses = []
    for step in range(3):                              # step 0 builds the arena (non-overlapped), steps 1-2 use the hooks
        loss = dp(X[step % 2, rank]).pow(2).mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert dp._reducer is not None and len(dp._reducer.buckets) > 1
    assert net.unused.weight.grad is None
    out = [p.detach().clone() for p in net.parameters()]
    if rank == 0:
        q.put(([w.numpy() for w in w0], [o.numpy() for o in out], X.numpy()))
    # replicas stay bit-identical
    chk = torch.cat([o.reshape(-1) for o in out])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    assert torch.equal(both[0], both[1])
    dist.destroy_process_group()


def test_data_parallel_equals_single_process_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    w0, out, X = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process restatement: average of the two ranks' gradients each step (DDP semantics)
    net = Net()
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(torch.from_numpy(w))
    X = torch.from_numpy(X)
    for step in range(3):
        grads = []
        for rank in range(2):
            net.zero_grad()
            net(X[step % 2, rank]).pow(2).mean().backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
        with torch.no_grad():
            for p, g0, g1 in zip(net.parameters(), *grads):
                if g0 is not None:
                    p.add_((g0 + g1) * 0.5, alpha=-0.1)
    for p, o in zip(net.parameters(), out):
        assert torch.allclose(p.detach(), torch.from_numpy(o), atol=1e-6, rtol=1e-5)


def test_world_size_one_is_a_passthrough():
    from ytvln import distributed as D
    net = Net()
    assert D.wrap_distributed_model(net, -1) is net
    dp = D.DataParallel(net)
    opt = ArenaSGD(net.parameters(), 0.1)
    dp.attach(opt)
    assert opt.grad_scale == 1.0
    dp(torch.randn(4, 6)).sum().backward()
    opt.step()


def test_rank_helpers_read_the_environment(monkeypatch):
    from ytvln import distributed as D
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SLURM_PROCID", "NODE_RANK", "SLURM_NTASKS"):
        monkeypatch.delenv(k, raising=False)
    assert (D.get_rank(), D.get_world_size(), D.get_local_rank()) == (0, 1, -1)
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("LOCAL_RANK", "3")
    assert (D.get_rank(), D.get_world_size(), D.get_local_rank()) == (3, 8, 3)
