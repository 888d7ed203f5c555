"""CPU, world_size = 2 (and 8) over gloo: the data-parallel gradient exchange (ytvln/distributed.py) is correct by construction.

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
    assert all(torch.equal(gathered[0], g_) for g_ in gathered), "parameters must be identical after wrapping"
    opt = ArenaSGD(net.parameters(), lr=0.1)
    dp.attach(opt)
    assert opt.grad_scale == 1.0 / world
    g = torch.Generator().manual_seed(7)
    X = torch.randn(2, world, 5, 6, generator=g)           # [step, rank, batch, features]
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
    assert all(torch.equal(both[0], b_) for b_ in both)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_data_parallel_equals_single_process_average(world):
    """world 8 = the node size of BASELINE configs[2]: buckets, the rank-0 broadcast, 1/world averaging and bit-identical replicas."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    w0, out, X = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # single-process restatement: average of the two ranks' gradients each step (DDP semantics)
    net = Net()
    with torch.no_grad():
        for p, w in zip(net.parameters(), w0):
            p.copy_(torch.from_numpy(w))
    X = torch.from_numpy(X)
    for step in range(3):
        grads = []
        for rank in range(world):
            net.zero_grad()
            net(X[step % 2, rank]).pow(2).mean().backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
        with torch.no_grad():
            for p, *gs in zip(net.parameters(), *grads):
                if gs[0] is not None:
                    p.add_(sum(gs) / world, alpha=-0.1)
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


def accum_worker(rank, world, port, q):
    """Two backward passes per optimizer step: the second one must not land in already-exchanged buckets unnoticed."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ytvln import distributed as D
    D.init_distributed(backend="gloo")
    torch.manual_seed(5)
    net = Net()
    dp = D.DataParallel(net, bucket_bytes=256)
    opt = ArenaSGD(net.parameters(), lr=0.1)
    dp.attach(opt)
    g = torch.Generator().manual_seed(11 + rank)
    xs = [torch.randn(5, 6, generator=g) for _ in range(6)]
    dp(xs[0]).pow(2).mean().backward(); opt.step(); opt.zero_grad()          # step 0 builds arena + reducer
    # (a) the documented way: all but the last backward under no_sync()
    with dp.no_sync():
        dp(xs[1]).pow(2).mean().backward()
    dp(xs[2]).pow(2).mean().backward()
    opt.step(); opt.zero_grad()
    chk = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    ok_sync = torch.equal(both[0], both[1])
    # (b) forgetting it: buckets get exchanged during the first backward; the second backward's gradients arrive late -> loud error
    dp(xs[3]).pow(2).mean().backward()
    dp(xs[4]).pow(2).mean().backward()
    raised = False
    try:
        opt.step()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    q.put((rank, ok_sync, raised))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_contract_no_sync_or_loud_error():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, ok_sync, raised in res:
        assert ok_sync, "replicas diverged under no_sync() accumulation"
        assert raised, "a late gradient in an already-exchanged bucket must raise"


def test_rccl_binding_resolves_pytorchs_librccl_without_a_gpu():
    """Host-only entry points of the RCCL binding: dlopen picks the librccl PyTorch already mapped; a unique id can be drawn."""
    import ctypes
    from ytvln import _lib
    from ytvln.distributed import RcclCommunicator, default_collective
    lib = _lib.load()
    _lib.call("ytvln_rccl_load", None)
    path = lib.ytvln_rccl_library_path().decode()
    assert os.path.realpath(path) == os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    v = ctypes.c_int()
    _lib.call("ytvln_rccl_version", ctypes.byref(v))
    assert v.value >= 21000
    uid = RcclCommunicator.new_unique_id()
    assert len(uid) == _lib.RCCL_UNIQUE_ID_BYTES
    with pytest.raises(RuntimeError, match="128"):
        _lib.call("ytvln_rccl_unique_id", ctypes.create_string_buffer(64), 64)
    with pytest.raises(RuntimeError, match="no communicator"):
        _lib.call("ytvln_rccl_allreduce", None, None, 0, 0, 0, None)
    assert lib.ytvln_rccl_destroy(None) == 0
    assert default_collective() == ("rccl" if torch.cuda.is_available() else "torch")


def digest_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ytvln import distributed as D
    D.init_distributed(backend="gloo")

    class Opt:
        def flat_grad(self):
            return torch.zeros(1000)
    gs = object.__new__(D.GraphedTrainStep)             # (capturing needs a device; the agreement check does not)
    gs.mode, gs.world, gs.group, gs.opt = "phased", world, None, Opt()
    gs._slices = [(0, 1000)]
    gs._group_slices = [[(0, 600)], [(600, 1000)]]
    gs.verify_layout_across_ranks()                      # identical on every rank: passes
    same = gs.layout_digest()
    if rank == world - 1:
        gs._group_slices = [[(0, 500)], [(500, 1000)]]   # one rank observed another phase boundary
    raised = False
    try:
        gs.verify_layout_across_ranks()
    except RuntimeError as e:
        raised = "different gradient-exchange layouts" in str(e)
    q.put((rank, raised, same))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_step_refuses_mismatched_exchange_layouts():
    """the phased step derives its exchange groups from what each rank observed during capture; ranks that disagree must get an
    error on EVERY rank before the first grouped all-reduce, not a hang inside RCCL."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=digest_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(raised for _, raised, _ in res)
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("n", [4, 8])
def test_bench_self_launch_command(monkeypatch, n):
    """`python bench.py --gpus N` without WORLD_SIZE re-execs under torch.distributed.run on 127.0.0.1 (README.md:98-100 counterpart)."""
    import importlib
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(bench.os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv) or (_ for _ in ()).throw(SystemExit(0)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    argv = seen["argv"]
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and f"--nproc-per-node={n}" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    assert argv[-6:] == ["--gpus", str(n), "--steps", "2", "--warmup", "1"] and argv[-7].endswith("bench.py")
