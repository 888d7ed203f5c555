"""GPU: the data-parallel path end to end on device tensors -- 2 ranks sharing the one GPU of the test box over gloo
(RCCL refuses duplicate devices; the code path -- AdamW arena, post-accumulate hooks, bucketed all-reduce, grad_scale --
is the one `bench.py --gpus N` runs over RCCL).  Checks replica equality and equality with a single-process step on the
averaged loss (DDP semantics, utils/distributed.py:99)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import ZERO_DROP, args_ns, cfg_dict

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(dev):
    from ytvln import synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    cfg = BertConfig(**cfg_dict("micro.json", **ZERO_DROP))
    cfg.args = args
    model = Lily(cfg, dropout_prob=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, 3).items()})
    return model.to(dev).train(), args


def _batch(rank, dev):
    from ytvln import synth
    return synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=40 + rank, ignore_rank_frac=0.0), dev)


def _worker(rank, world, port, q, mode="eager"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))
    import torch.distributed as dist
    from ytvln import distributed as D, utils_init as U
    from ytvln.vilbert_init import get_optimization
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    D.init_distributed(backend="gloo")
    model, args = _build(dev)
    args.learning_rate = 1e-3
    # collective="torch": two ranks share ONE device here, which RCCL refuses (the RCCL data plane is covered by test_rccl_gpu.py)
    dp = D.DataParallel(model, bucket_bytes=64 << 10, collective="torch")          # small buckets -> several overlapped collectives
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    dp.attach(opt)
    batch = _batch(rank, dev)
    if mode == "accum":      # 2 micro-steps per optimizer step: the exchange happens once, on the accumulated gradients
        args.gradient_accumulation_steps = 2
        batch2 = _batch(rank + 2, dev)
        for step in range(6):
            U.train_step(dp, opt, sched, batch if step % 2 == 0 else batch2, args, step, all_options=True)
    elif mode == "eager":
        for step in range(3):
            U.train_step(dp, opt, sched, batch, args, step, all_options=True)
    else:       # one eager step, then two replays of the two-graph step with the exchange between the graphs
        U.train_step(dp, opt, sched, batch, args, 0, all_options=True)
        if mode == "phased":
            os.environ["YTVLN_DP_CUTS"] = "t0,c0,v1"
        gs = D.GraphedTrainStep(dp, opt, lambda backward=None: U.train_step(dp, opt, None, batch, args, 0, all_options=True, optimizer_step=False,
                                                                            backward=backward)[0],
                                bucket_bytes=64 << 10, mode="phased" if mode == "phased" else None)
        assert len(gs._slices) > 1 and gs.mode == ("phased" if mode == "phased" else "split")
        if mode == "phased":
            # every parameter that receives a gradient travels in exactly one group (a parameter left out would make the replicas diverge
            # and the comparison with the single-process average below fail)
            owned = sum(hi - lo for g in gs._group_slices for lo, hi in g)
            assert len(gs.graphs) == 4 and owned == sum(opt.arena_range(p)[1] for p in model.parameters() if opt.arena_range(p) is not None)
        for step in range(2):
            loss = gs.step(sched)
        assert torch.isfinite(loss).item()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    assert dp._reducer is not None and len(dp._reducer.buckets) > 1
    assert all(opt.state[p]["step"] == 3 for p in model.parameters() if p in opt.state and "step" in opt.state[p])
    if rank == 0:
        q.put(flat.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["eager", "graphed", "phased", "accum"])
def test_two_ranks_match_single_process_average(dev, lib, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process: loss = mean of the two ranks' losses  <=>  averaged gradients
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    model, args = _build(dev)
    args.learning_rate = 1e-3
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    batches = [_batch(r, dev) for r in range(2)]
    if mode == "accum":       # per optimizer step every rank sees (batch_r, batch_{r+2}), each micro-loss / 2, ranks averaged
        batches = [_batch(r, dev) for r in range(4)]
    for step in range(3):
        total = None
        for b in batches:
            outputs = model(*U.get_model_input(b, all_options=True))
            for task, flag in U.TASKS:
                _, _, l, _ = U.get_loss_correct(b, outputs, task, args, None, True, all_options=True)
                l = (1.0 / len(batches)) * (args.traj_loss_scale * l if task == "traj" else l)
                total = l if total is None else total + l
        total.backward()
        opt.step(); sched.step(); opt.zero_grad()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()
    assert np.allclose(got, ref, atol=2e-6, rtol=2e-5), float(np.abs(got - ref).max())
