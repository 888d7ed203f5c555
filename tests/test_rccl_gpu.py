"""GPU: the RCCL data plane of the data-parallel path on the ONE device of the test box.

RCCL refuses two ranks on one device, but a one-rank communicator is legal (`ncclCommInitRank(nranks=1)`): every collective
really goes through librccl (kernel launch on the given stream, graph capture, group calls) and is the identity on the data,
so a data-parallel run must equal the plain single-process run BIT FOR BIT.  Covered here:
  * the `ytvln_rccl_*` C ABI itself (load / version / unique id / init / all-reduce / grouped slices / broadcast / destroy);
  * `DataParallel` + bucket hooks on a comm stream, eager; `GraphedTrainStep` in both modes (exchange between two graphs; exchange
    captured INTO the graph) -- over the C ABI communicator;
  * the same wrapper over torch.distributed's "nccl" (= RCCL) backend with its watchdog thread alive during capture;
  * `bench.py --gpus N` starting its own ranks (2 ranks share the GPU over gloo; utils/distributed.py:63-104, README.md:98-100).
Each case runs in a spawned process (process-group and communicator state never leak into the other GPU tests)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT
from helpers import ZERO_DROP, args_ns, cfg_dict

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# the bf16-resident attention kernels exist for head dimensions 64 / 128: the micro config widened to two heads of 64 in every stream
_WIDE = dict(hidden_size=128, num_attention_heads=2, intermediate_size=128, v_hidden_size=128, v_num_attention_heads=2, v_intermediate_size=128,
             bi_hidden_size=128, bi_num_attention_heads=2)


def _build(dev, wide=False):
    from ytvln import synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    cfg = BertConfig(**cfg_dict("micro.json", **ZERO_DROP, **(_WIDE if wide else {})))
    cfg.args = args
    model = Lily(cfg, dropout_prob=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, 3).items()})
    return model.to(dev).train(), args


def _batch(dev, seed=40):
    from ytvln import synth
    return synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=seed, ignore_rank_frac=0.0), dev)


def _flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def _plain_run(dev, steps=3, precision="fp32", wide=False):
    from ytvln import ops, utils_init as U
    from ytvln.vilbert_init import get_optimization
    ops.set_matmul_precision(precision)
    try:
        return _plain_run_body(dev, steps, U, get_optimization, precision != "fp32" or wide)
    finally:
        ops.set_matmul_precision("fp32")


def _plain_run_body(dev, steps, U, get_optimization, wide):
    model, args = _build(dev, wide)
    args.learning_rate = 1e-3
    opt, sched, _, _ = get_optimization(args, model, 10 if steps <= 3 else steps + 5, None)
    batch = _batch(dev)
    for step in range(steps):
        U.train_step(model, opt, sched, batch, args, step, all_options=True)
    torch.cuda.synchronize()
    return _flat(model)


def _worker(case, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))
        import torch.distributed as dist
        from ytvln import distributed as D, utils_init as U
        from ytvln.vilbert_init import get_optimization
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        case, _, precision = case.partition("@")          # "...@bf16": the bf16-resident path (bf16 weight arena written by AdamW, bf16 activations)
        from ytvln import ops
        ops.set_matmul_precision(precision or "fp32")
        collective, mode = case.split(":")
        nreplay = 2
        if "#" in mode:                           # "phased#50": that many replays (soak of the phased exchange: comm stream, per-group AdamW)
            mode, _, nr = mode.partition("#")
            nreplay = int(nr)
        metrics = mode.endswith("+metrics")       # the reference's default: local_rank != -1 and --skip_all_reduce off -> logged metrics are all-reduced
        mode = mode.replace("+metrics", "")
        if mode == "phased3":           # explicit cut list: a text-only cut below the co-attention layer, the layer itself, an image-only cut above
            os.environ["YTVLN_DP_CUTS"] = "t0,c0,v1"
            mode = "phased"
        D.init_distributed(backend="nccl" if collective == "torch" else "gloo", force=True)
        assert dist.get_world_size() == 1 and dist.get_backend() == ("nccl" if collective == "torch" else "gloo")
        model, args = _build(dev, wide=bool(precision))
        args.learning_rate = 1e-3
        if metrics:
            args.local_rank, args.skip_all_reduce = 0, False
        dp = D.DataParallel(model, bucket_bytes=64 << 10, collective=collective, always_exchange=True)
        assert (dp.comm is not None) == (collective == "rccl")
        info = {}
        if dp.comm is not None:
            info["library"] = dp.comm.library
        opt, sched, _, _ = get_optimization(args, model, 10 if nreplay <= 2 else nreplay + 6, None)     # (the schedule of _plain_run_body)
        dp.attach(opt)
        batch = _batch(dev)
        if mode == "eager":
            for step in range(3):
                U.train_step(dp, opt, sched, batch, args, step, all_options=True)
            assert dp._reducer is not None and len(dp._reducer.buckets) > 1
            assert dp._reducer.collectives == 3 * len(dp._reducer.buckets)      # every bucket, every step
            info["collectives"] = dp._reducer.collectives
        else:
            U.train_step(dp, opt, sched, batch, args, 0, all_options=True)
            gs = D.GraphedTrainStep(dp, opt, lambda backward=None: U.train_step(dp, opt, None, batch, args, 0, all_options=True,
                                                                                optimizer_step=False, backward=backward)[0],
                                    bucket_bytes=64 << 10, mode=mode)
            assert gs.exchange and len(gs._slices) > 1
            if mode == "phased":
                info["phases"] = len(gs.graphs)
                info["groups"] = [sum(hi - lo for lo, hi in g) for g in gs._group_slices]
                info["arena"] = int(opt.flat_grad().numel())
                info["params"] = sum(p.numel() for p in model.parameters() if p.grad is not None or True)
            mem = []
            for step in range(nreplay):
                loss = gs.step(sched)
                if step in (4, nreplay - 1):
                    torch.cuda.synchronize()
                    mem.append((torch.cuda.memory_allocated(), torch.cuda.memory_reserved()))
            assert torch.isfinite(loss).item()
            if nreplay > 5:
                assert mem[0] == mem[-1], f"device memory moved during the replays: {mem}"          # constant footprint over the soak
                info["mem"] = mem
            if mode == "phased":        # the cut points are gone once the phases are recorded: an eager step sees the uncut graph
                assert all(not m.cut_after for m in model.modules() if hasattr(m, "cut_after"))
        torch.cuda.synchronize()
        if metrics:       # the metric reductions ride on the RCCL data plane (stream-ordered, capturable), not on the gloo control plane
            red = {"loss": {}, "accuracy": {}}
            out = dp(*U.get_model_input(batch, True))
            l = U.compute_metrics_independent(batch, out, "ranking", args, None, red, all_options=True)
            assert torch.allclose(red["loss"]["ranking"], l.detach()) and red["loss"]["ranking"].is_cuda
            assert D.metrics_world_size() == 1
        if dp.comm is not None:
            dp.comm.check_async_error()
        got = _flat(model)
        dp.close()
        dist.destroy_process_group()
        q.put(("ok", got, info))
    except Exception as e:      # surface the failure in the parent instead of a bare exit code
        import traceback
        q.put(("error", traceback.format_exc(), {}))
        raise e


@pytest.mark.parametrize("case", ["rccl:eager", "rccl:split", "rccl:single", "rccl:phased", "rccl:phased3", "rccl:phased#50", "rccl:phased3#50", "torch:eager", "torch:split",
                                  "rccl:eager+metrics", "rccl:split+metrics", "rccl:phased+metrics", "rccl:eager@bf16", "rccl:phased@bf16"])
def test_one_rank_rccl_world_equals_plain_run(dev, lib, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(case, _free_port(), q))
    p.start()
    status, got, info = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", got
    assert p.exitcode == 0
    case, _, precision = case.partition("@")
    nreplay = int(case.partition("#")[2] or 2)
    case = case.partition("#")[0]
    ref = _plain_run(dev, steps=1 + nreplay, precision=precision or "fp32")
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())     # identity exchange, grad_scale 1: bit-identical
    if precision:
        assert not np.array_equal(ref, _plain_run(dev, wide=True)), "the bf16-resident run must differ from the fp32 one"
    if case.startswith("rccl"):
        assert "rccl" in os.path.basename(info["library"])
    if case == "rccl:phased3":
        assert info["phases"] == 4, info
    if case in ("rccl:phased", "rccl:phased3"):
        # micro config: 2 + 2 layers, one co-attention layer -> cuts after c0 (and t1 when the text block below it is that long); every
        # group but the last is exchanged under the following phases, and together the groups cover every parameter that has a gradient
        assert info["phases"] >= 2 and len(info["groups"]) == info["phases"]
        assert all(g > 0 for g in info["groups"][:1]) and info["groups"][-1] > 0
        assert sum(info["groups"]) <= info["arena"] and sum(info["groups"]) >= 0.95 * info["arena"], info


def test_rccl_c_abi_collectives(dev, lib):
    """The binding itself: a one-rank communicator, every entry point, results checked on the device."""
    import ctypes
    from ytvln import _lib
    from ytvln.distributed import RcclCommunicator
    v = ctypes.c_int()
    _lib.call("ytvln_rccl_version", ctypes.byref(v))
    assert v.value >= 21000, v.value
    path = lib.ytvln_rccl_library_path().decode()
    # the copy already mapped by PyTorch is the one that must be bound (same HIP runtime as torch's streams)
    assert os.path.realpath(path) == os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")), path
    uid = RcclCommunicator.new_unique_id()
    assert len(uid) == 128 and uid != RcclCommunicator.new_unique_id()
    comm = RcclCommunicator(0, 1, dev, uid)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(1 << 20, generator=g).to(dev)
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce(x, "sum", stream=side)                      # on a non-default stream
    torch.cuda.current_stream().wait_stream(side)
    comm.all_reduce(x, "max")
    comm.all_reduce_slices(x, [(0, 1000), (1000, 70000), (70000, x.numel())])
    comm.broadcast(x, root=0)
    i = torch.arange(17, device=dev)
    comm.all_reduce(i, "sum")
    d = torch.full((5,), 1.5, dtype=torch.float64, device=dev)
    comm.all_reduce(d, "min")
    torch.cuda.synchronize()
    comm.check_async_error()
    assert torch.equal(x, ref) and torch.equal(i, torch.arange(17, device=dev)) and float(d.sum()) == 7.5
    with pytest.raises(RuntimeError, match="contiguous"):
        comm.all_reduce(x.view(1024, 1024).t())
    with pytest.raises(RuntimeError, match="lives on"):
        comm.all_reduce(torch.zeros(4))
    with pytest.raises(RuntimeError, match="ytvln_rccl_init failed"):
        RcclCommunicator(3, 2, dev, uid)                        # rank outside the world: rejected before RCCL is touched
    comm.close()
    comm.close()                                                # idempotent


def _run_bench(extra, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1_tiny_mlm_bs2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-variants", *extra], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(dev, lib):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: re-execs under torch.distributed.run, rank 0 prints ONE line.
    (2 ranks on the single test GPU need the gloo exchange; on an N-GPU node the same command runs the RCCL communicator.)"""
    out = _run_bench(["--gpus", "2"], {"YTVLN_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_pairs"] == 2 * out["config"]["pairs_per_gpu"]
    assert out["config"]["replica_checksum_spread"] == 0.0          # two ranks, different data, identical parameters after the steps
    assert np.isfinite(out["final_loss"]) and out["value"] > 0
    assert "two hipGraphs" in out["config"]["execution"] or "eager" in out["config"]["execution"]      # gloo data plane: the split form


def test_bench_dp_path_over_the_c_abi_communicator(dev, lib):
    """The N > 1 code path of bench.py (DataParallel, two-graph step, arena all-reduce through ytvln_rccl_*) in a one-rank world."""
    out = _run_bench(["--gpus", "1", "--dp-selftest"], {})
    assert out["n_gpus"] == 1 and out["config"]["dp_selftest"] is True
    assert out["config"]["gradient_exchange"].startswith("ytvln_rccl_")
    assert "hipGraphs per step" in out["config"]["execution"] and "[phased;" in out["config"]["execution"], out["config"]["execution"]
    assert np.isfinite(out["final_loss"])
