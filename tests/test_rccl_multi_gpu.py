"""GPU, TWO OR MORE devices: the data-parallel gradient exchange really crossing devices over RCCL / xGMI (BASELINE configs[2]; the
reference's path is DDP over NCCL, utils/distributed.py:63-104).

Every other N > 1 test in this tree is gloo (CPU, or two ranks sharing the one GPU of the test box) or a one-rank RCCL world in which
all-reduce is the identity.  This file runs one rank per device through the C ABI's own communicator (`ytvln_rccl_*`) and SKIPS on a
one-device box (`torch.cuda.device_count() < 2`): it lights up the moment a multi-GPU node runs `pytest -m gpu`.

  * `ytvln_rccl_allreduce_slices_f32` / `ytvln_rccl_allreduce` of DISTINCT per-rank arenas equal the fp64 sum of what every rank held;
  * `DataParallel` eager bucket hooks, `GraphedTrainStep` "split" and "phased": steps on DIFFERENT per-rank batches equal the
    single-process step on the averaged loss (the bar of test_dp_gpu.py::test_two_ranks_match_single_process_average) and every replica
    holds the same bits afterwards (checksum spread == 0.0).
"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_dp_gpu import _batch, _build, _free_port

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                                 reason="needs >= 2 HIP devices: one rank per device over RCCL (the 1-GPU test box has one)")]


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))


def _allreduce_worker(rank, world, port, q):
    _env(rank, world, port)
    import torch.distributed as dist
    from ytvln import distributed as D
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    D.init_distributed(backend="gloo")                       # control plane only: carries the RCCL unique id
    comm = D.RcclCommunicator.from_process_group(dev)
    n = (1 << 22) + 12345                                    # 16 MiB + an odd tail
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(n, generator=g)
    flat = mine.to(dev)
    slices = [(0, 1000), (1000, 1 << 20), (1 << 20, (1 << 21) + 7), ((1 << 21) + 7, n)]
    comm.all_reduce_slices(flat, slices)
    whole = mine.to(dev)
    comm.all_reduce(whole)
    torch.cuda.synchronize(dev)
    # every rank rebuilds every rank's input from its seed: the expected sum in fp64
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)).double() for r in range(world))
    e1 = float((flat.cpu().double() - ref).abs().max())
    e2 = float((whole.cpu().double() - ref).abs().max())
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, e1, e2, float(ref.abs().max())))


def _dp_worker(rank, world, port, q, mode):
    _env(rank, world, port)
    import torch.distributed as dist
    from ytvln import distributed as D, utils_init as U
    from ytvln.vilbert_init import get_optimization
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    D.init_distributed(backend="gloo")
    model, args = _build(dev)
    args.learning_rate = 1e-3
    dp = D.DataParallel(model, bucket_bytes=64 << 10, collective="rccl")
    assert dp.collective == "rccl" and dp.comm is not None and dp.comm.world == world
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    dp.attach(opt)
    batch = _batch(rank, dev)
    if mode == "eager":
        for step in range(3):
            U.train_step(dp, opt, sched, batch, args, step, all_options=True)
    else:
        U.train_step(dp, opt, sched, batch, args, 0, all_options=True)
        if mode == "phased":
            os.environ["YTVLN_DP_CUTS"] = "t0,c0,v1"
        gs = D.GraphedTrainStep(dp, opt, lambda backward=None: U.train_step(dp, opt, None, batch, args, 0, all_options=True, optimizer_step=False,
                                                                            backward=backward)[0],
                                bucket_bytes=64 << 10, mode="phased" if mode == "phased" else None)
        for step in range(2):
            loss = gs.step(sched)
        assert torch.isfinite(loss).item()
    torch.cuda.synchronize(dev)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    sums = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(sums, flat.double().sum().reshape(1))
    spread = float(max(sums) - min(sums))
    alls = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(alls, flat)
    same = all(torch.equal(alls[0], a) for a in alls[1:])
    if dp.comm is not None:
        dp.comm.close()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, spread, same, flat.numpy() if rank == 0 else None))


def _spawn(target, world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.timeout(900)
def test_rccl_allreduce_sums_distinct_arenas_across_devices():
    world = min(torch.cuda.device_count(), 8)
    for rank, e1, e2, scale in _spawn(_allreduce_worker, world):
        bar = 4e-7 * scale * world + 1e-6           # fp32 ring sums of `world` terms against the fp64 sum
        assert e1 <= bar and e2 <= bar, (rank, e1, e2, bar)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["eager", "split", "phased"])
def test_two_devices_match_single_process_average(dev, lib, mode):
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    res = _spawn(_dp_worker, 2, mode)
    assert all(spread == 0.0 and same for _, spread, same, _ in res), [(r, s, e) for r, s, e, _ in res]      # replica_checksum_spread == 0.0
    got = res[0][3]
    model, args = _build(dev)
    args.learning_rate = 1e-3
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    batches = [_batch(r, dev) for r in range(2)]
    for step in range(3):
        total = None
        for b in batches:
            outputs = model(*U.get_model_input(b, all_options=True))
            for task, flag in U.TASKS:
                _, _, l, _ = U.get_loss_correct(b, outputs, task, args, None, True, all_options=True)
                l = 0.5 * (args.traj_loss_scale * l if task == "traj" else l)
                total = l if total is None else total + l
        total.backward()
        opt.step(); sched.step(); opt.zero_grad()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    assert np.allclose(got, ref, atol=2e-6, rtol=2e-5), float(np.abs(got - ref).max())
