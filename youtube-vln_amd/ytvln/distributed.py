"""Data-parallel training over RCCL/xGMI: one process per GPU, flat-bucket gradient all-reduce overlapped with backward.

Counterpart of the reference's `utils/distributed.py:63-153` (`init_distributed`, `set_cuda`, `wrap_distributed_model` =
`DistributedDataParallel(find_unused_parameters=True)`).  `torch.distributed`'s "nccl" backend IS RCCL on ROCm; the
design choices here are for MI355X's point-to-point xGMI fabric rather than a translation of DDP's defaults:

  * gradients already live in ONE flat fp32 arena (ytvln.optimization.AdamW), so a bucket is a contiguous slice of it --
    no flatten/unflatten copies, and few LARGE collectives (default 256 MiB buckets; DDP's 25 MB buckets would put ~40
    launch latencies on a per-link-bound ring);
  * the never-used tensors (q_dense*, bi_seq_relationship, unflagged heads) are simply absent from the
    arena, so no `find_unused_parameters` graph walk is needed and the bucket layout is identical on all ranks;
  * buckets are all-reduced (SUM) as soon as their last gradient has been accumulated during backward (post-accumulate
    hooks), on RCCL's own stream; the 1/world_size averaging is folded into the fused AdamW kernel (`grad_scale`).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn


def get_rank(default: int = 0) -> int:
    for k in ("RANK", "SLURM_PROCID", "NODE_RANK"):
        if k in os.environ:
            return int(os.environ[k])
    return default


def get_world_size(default: int = 1) -> int:
    for k in ("WORLD_SIZE", "SLURM_NTASKS"):
        if k in os.environ:
            return int(os.environ[k])
    return default


def get_local_rank(args=None) -> int:
    if "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    return getattr(args, "local_rank", -1) if args is not None else -1


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int]:
    """env:// rendezvous (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE), like utils/distributed.py:63-90."""
    world = get_world_size()
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (the only mode this driver supports)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, init_method="env://", rank=get_rank(), world_size=world)
    return dist.get_rank(), dist.get_world_size()


def set_cuda(args=None):
    """(default_gpu, n_gpu, device) -- utils/distributed.py:122-153."""
    local_rank = get_local_rank(args)
    if not torch.cuda.is_available():
        raise RuntimeError("ytvln needs a HIP device; there is no CPU training path")
    if local_rank == -1:
        return True, torch.cuda.device_count(), torch.device("cuda", 0)
    torch.cuda.set_device(local_rank)
    rank, _ = init_distributed()
    return rank == 0, 1, torch.device("cuda", local_rank)


class GradBucketReducer:
    """All-reduces contiguous slices ("buckets") of a flat gradient tensor as they become ready during backward.

    layout: [(param, offset, numel)] in arena order.  Device-agnostic (tested on CPU with gloo, world_size 2)."""

    def __init__(self, flat: torch.Tensor, layout: Sequence[Tuple[torch.nn.Parameter, int, int]], bucket_bytes: int = 256 << 20,
                 group=None, overlap: bool = True, enabled_fn: Optional[Callable[[], bool]] = None):
        self.flat, self.group, self.overlap = flat, group, overlap
        self.enabled_fn = enabled_fn or (lambda: True)
        self._offsets = {id(p): off for p, off, _ in layout}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        cap = max(1, bucket_bytes // flat.element_size())
        self.buckets: List[dict] = []
        cur = None
        for p, off, n in layout:
            if cur is None or (off + n - cur["lo"]) > cap and cur["params"]:
                cur = dict(lo=off, hi=off, params=[], pending=0, handle=None, launched=False)
                self.buckets.append(cur)
            cur["params"].append(p)
            cur["hi"] = max(cur["hi"], off + n)
        self._of = {}
        for b in self.buckets:
            for p in b["params"]:
                self._of[id(p)] = b
        self._hooks = []
        self.reset()
        if overlap and self.world > 1:
            for p, _, _ in layout:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def reset(self):
        for b in self.buckets:
            b["pending"], b["handle"], b["launched"] = len(b["params"]), None, False

    def _launch(self, b):
        b["launched"] = True
        if self.world > 1:
            b["handle"] = dist.all_reduce(self.flat[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        if not self.enabled_fn():
            return
        b = self._of.get(id(p))
        if b is None or b["launched"]:
            return
        if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + self.flat.element_size() * self._offsets[id(p)]:
            return      # gradient is not (yet) a view of the arena: finish() reduces the bucket after re-adoption
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self):
        """Called right before the optimizer consumes the gradients: launch what is missing, wait for everything."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class DataParallel(nn.Module):
    """`wrap_distributed_model` counterpart: replicas + gradient averaging.  Use `attach(optimizer)` once."""

    def __init__(self, module: nn.Module, bucket_bytes: int = 256 << 20, group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self.group, self.bucket_bytes = group, bucket_bytes
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._reducer: Optional[GradBucketReducer] = None
        self._opt = None
        self.require_backward_grad_sync = True
        if broadcast and self.world > 1:
            with torch.no_grad():                       # DDP broadcasts rank-0 weights at wrap time
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=group)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def zero_grad(self, set_to_none: bool = False):
        if self._opt is not None:
            self._opt.zero_grad()
        else:
            self.module.zero_grad(set_to_none=True)

    def attach(self, optimizer):
        """Hook the gradient exchange into `optimizer.step()` (ytvln.optimization.AdamW calls `grad_sync(flat, layout)`
        after (re)adopting gradients into its arena and before the fused update)."""
        self._opt = optimizer
        optimizer.grad_scale = 1.0 / self.world
        optimizer.grad_sync = self._sync
        return self

    def _sync(self, flat: torch.Tensor, layout):
        if self.world == 1:
            return
        if self._reducer is None or self._reducer.flat.data_ptr() != flat.data_ptr():
            if self._reducer is not None:
                self._reducer.remove()
            self._reducer = GradBucketReducer(flat, layout, self.bucket_bytes, self.group,
                                              enabled_fn=lambda: self.require_backward_grad_sync)
        self._reducer.finish()


class GraphedTrainStep:
    """A training step as TWO hipGraphs with the gradient exchange between them:

        graph A   forward, losses, backward, adoption of the stray gradients into the flat arena
        eager     all-reduce(SUM) of the arena in `bucket_bytes` slices over RCCL   (world > 1; never captured)
        graph B   fused AdamW (1/world folded in) 

    The host enqueues two graph launches and a handful of collectives per step instead of ~1500 kernels, so N processes do
    not compete for host cores; RCCL calls stay ordinary stream work, exactly as in the eager path.  (The eager path overlaps
    the exchange with backward; here it follows backward -- 1 GB over xGMI, a few ms against a >100 ms step.)

    `fwd_bwd()` must run forward + backward only (utils_init.train_step(..., optimizer_step=False)) on STATIC input tensors and
    return the loss tensor; refill the inputs in place between steps.  Run >= 1 eager step first (arenas, allocator warm-up)."""

    def __init__(self, model: nn.Module, optimizer, fwd_bwd: Callable[[], torch.Tensor], bucket_bytes: int = 256 << 20, group=None):
        self.opt, self.group, self.bucket_bytes = optimizer, group, bucket_bytes
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dp = model if isinstance(model, DataParallel) else None
        if optimizer.flat_grad() is None:
            raise RuntimeError("run at least one eager training step before capturing")
        optimizer.zero_grad()
        optimizer.grad_scale = 1.0 / self.world
        torch.cuda.synchronize()
        if dp is not None:
            dp.require_backward_grad_sync = False       # the bucket hooks must not launch collectives into the capture
        try:
            # thread_local: RCCL's watchdog thread polls events while we capture; only this thread's calls belong to the graph
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
                self.loss = fwd_bwd()
                optimizer.capture_adopt()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
                optimizer.capture_update()
        finally:
            if dp is not None:
                dp.require_backward_grad_sync = True
        optimizer.zero_grad()
        torch.cuda.synchronize()
        flat = optimizer.flat_grad()
        cap = max(1, bucket_bytes // flat.element_size())
        self._slices = [(lo, min(lo + cap, flat.numel())) for lo in range(0, flat.numel(), cap)]

    def step(self, scheduler=None) -> torch.Tensor:
        self.graph_a.replay()
        if self.world > 1:
            flat = self.opt.flat_grad()
            for lo, hi in self._slices:
                dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self.opt.prepare_replay()
        self.graph_b.replay()
        if scheduler is not None:
            scheduler.step()
        return self.loss


def wrap_distributed_model(model: nn.Module, local_rank: int = -1, **kw) -> nn.Module:
    """utils/distributed.py:97-104."""
    if local_rank != -1 and dist.is_initialized() and dist.get_world_size() > 1:
        return DataParallel(model, **kw)
    return model
