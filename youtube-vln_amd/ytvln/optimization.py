"""Fused AdamW + LR schedules with the reference's semantics (`vilbert/optimization.py`).

`AdamW` keeps the reference constructor (`AdamW(params, lr, betas, eps, weight_decay, correct_bias)`), the per-parameter
state layout (`step`, `exp_avg`, `exp_avg_sq` -> checkpoints stay interchangeable) and its exact update rule
(`optimization.py:141-187`: denom = sqrt(v)+eps, bias correction folded into the step size, decoupled decay applied AFTER
the update, tensors without a gradient skipped entirely).  Execution differs: on the first `step()` the parameters that
received a gradient are moved into one flat fp32 arena (their `.data` / `.grad` become views), moments live in two more
arenas, and a whole step is one HIP kernel launch per (param-group, step-count) class instead of ~8 ATen kernels per
tensor x 541 tensors.  The flat gradient arena is also what the data-parallel all-reduce buckets (ytvln/distributed.py).
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

from . import ops


CHUNK = 16384       # elements per workgroup of the fused kernel
ALIGN = 4           # arena offsets are multiples of 4 floats (16-byte vector access)


class ConstantLRSchedule(LambdaLR):
    """optimization.py:26-30."""

    def __init__(self, optimizer, last_epoch=-1):
        super().__init__(optimizer, lambda _: 1.0, last_epoch=last_epoch)


class WarmupConstantSchedule(LambdaLR):
    """optimization.py:33-45."""

    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1.0, self.warmup_steps))
        return 1.0


class WarmupLinearSchedule(LambdaLR):
    """optimization.py:48-61: linear warm-up to 1 over `warmup_steps`, then linear decay to 0 at `t_total`."""

    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.t_total = t_total
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))


class WarmupCosineSchedule(LambdaLR):
    """optimization.py:64-82: linear warm-up, then 0.5 * (1 + cos(2 pi cycles progress)) over the remaining steps (half a period by default),
    floored at 0."""

    def __init__(self, optimizer, warmup_steps, t_total, cycles=0.5, last_epoch=-1):
        self.warmup_steps, self.t_total, self.cycles = warmup_steps, t_total, cycles
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1.0, self.warmup_steps))
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(2.0 * math.pi * float(self.cycles) * progress)))


class WarmupCosineWithHardRestartsSchedule(LambdaLR):
    """optimization.py:85-105: linear warm-up, then `cycles` cosine decays from 1 to 0 with hard restarts; 0 once the schedule is over."""

    def __init__(self, optimizer, warmup_steps, t_total, cycles=1.0, last_epoch=-1):
        self.warmup_steps, self.t_total, self.cycles = warmup_steps, t_total, cycles
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1, self.warmup_steps))
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(self.cycles) * progress) % 1.0))))


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._arena = None          # dict(p, g, m, v flat tensors; index: id(param) -> (offset, numel))
        self._launch = None         # list of launch classes
        self.grad_scale = 1.0       # multiplied into the gradient inside the kernel (set by data-parallel wrappers)

    # ---- arena management -----------------------------------------------------------------------------------------
    def _members(self):
        return [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]

    def flat_grads(self):
        """The flat gradient arena (None before the first step): what a data-parallel wrapper all-reduces."""
        return None if self._arena is None else self._arena["g"]

    def arena_layout(self) -> Dict[int, tuple]:
        return {} if self._arena is None else dict(self._arena["index"])

    def _build_arena(self, members):
        dev = members[0][1].device
        for _, p in members:
            if p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("ytvln AdamW needs fp32 parameters on a HIP device (no CPU fallback)")
            if p.grad.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
        old = self._arena
        index, off = {}, 0
        for _, p in members:
            index[id(p)] = (off, p.numel())
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        flat = {k: torch.zeros(off, dtype=torch.float32, device=dev) for k in ("p", "g", "m", "v")}
        with torch.no_grad():
            for _, p in members:
                o, n = index[id(p)]
                flat["p"][o:o + n].copy_(p.data.reshape(-1))
                flat["g"][o:o + n].copy_(p.grad.reshape(-1))
                st = self.state[p]
                if "exp_avg" in st:                      # carried over from a previous arena / a loaded checkpoint
                    flat["m"][o:o + n].copy_(st["exp_avg"].reshape(-1))
                    flat["v"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                else:
                    st["step"] = 0
                p.data = flat["p"][o:o + n].view(p.shape)
                p.grad = flat["g"][o:o + n].view(p.shape)
                st["exp_avg"] = flat["m"][o:o + n].view(p.shape)
                st["exp_avg_sq"] = flat["v"][o:o + n].view(p.shape)
        self._arena = dict(flat, index=index, ids=[id(p) for _, p in members], pb=None, pb_versions={})
        self._launch = None
        del old
        # let the weight-gradient GEMMs write straight into the gradient arena and the packed projections alias the
        # parameter arena: every member parameter carries its slot (ytvln.ops.ArenaSlot)
        self._written = set()
        import weakref
        me = weakref.ref(self)
        for _, p in members:
            o, n = index[id(p)]
            p._ytvln_slot = ops.ArenaSlot(flat["p"], flat["g"], o, n, self._written, me)

    def bf16_arena(self, params=None):
        """The bf16 copy of the parameter arena (bf16-resident path, ytvln.ops._bf16_weight): created on first use by one cast of the whole
        arena, from then on refreshed by the AdamW kernel itself (ytvln_adamw_f32_bf16copy).  `params`: the parameters about to be read --
        if torch modified one of them in place since the copy was made (load_state_dict, manual edits: their version counters moved), its
        slice is cast again.  None before the arena exists."""
        a = self._arena
        if a is None:
            return None
        if a["pb"] is None:
            with ops.TwoStream.shared_write():
                a["pb"] = torch.empty(a["p"].numel(), dtype=torch.bfloat16, device=a["p"].device)
                ops.call("ytvln_cast_f32_bf16", a["p"].data_ptr(), a["p"].numel(), 1, a["p"].numel(), a["pb"].data_ptr(), a["p"].numel(), ops._stream())
            a["pb_versions"] = {id(p): p._version for _, p in self._members()}
        vers = a["pb_versions"]
        for p in (params or ()):
            rng = a["index"].get(id(p))
            if rng is None:
                return None
            if vers.get(id(p)) != p._version:
                if id(p) in vers:          # modified behind the optimizer's back: refresh this slice
                    o, n = rng
                    with ops.TwoStream.shared_write():
                        ops.call("ytvln_cast_f32_bf16", a["p"].data_ptr() + 4 * o, n, 1, n, a["pb"].data_ptr() + 2 * o, n, ops._stream())
                vers[id(p)] = p._version
        return a["pb"]

    def _ensure_arena(self):
        members = self._members()
        if not members:
            return members
        if self._arena is None or self._arena["ids"] != [id(p) for _, p in members]:
            self._build_arena(members)
            return members
        fp, fg, index = self._arena["p"], self._arena["g"], self._arena["index"]
        base_p, base_g = fp.data_ptr(), fg.data_ptr()
        stale_data = False
        with torch.no_grad():
            for _, p in members:
                o, n = index[id(p)]
                if p.data_ptr() != base_p + 4 * o:
                    stale_data = True
                    break
                if p.grad.data_ptr() != base_g + 4 * o:       # e.g. model.zero_grad() dropped the views: re-adopt
                    fg[o:o + n].copy_(p.grad.reshape(-1))
                    p.grad = fg[o:o + n].view(p.shape)
        if stale_data:                                         # parameters were re-allocated (model.to(...)): rebuild
            self._build_arena(members)
        return members

    def _build_launch(self, members):
        classes: Dict[tuple, list] = {}
        for gi, p in members:
            classes.setdefault((gi, self.state[p]["step"]), []).append(p)
        index, dev = self._arena["index"], self._arena["p"].device
        launch = []
        for (gi, step), plist in classes.items():
            wd = float(self.param_groups[gi]["weight_decay"])
            rec = bytearray()
            n = 0
            for p in plist:
                o, numel = index[id(p)]
                for c in range(0, numel, CHUNK):
                    rec += struct.pack("<qqff", o + c, min(CHUNK, numel - c), wd, 0.0)
                    n += 1
            table = torch.frombuffer(rec, dtype=torch.uint8).to(dev)
            # hyper-parameters travel through a small ring of pinned host buffers (async H2D, no per-step stream sync);
            # an event per slot guards reuse should the host ever run a full ring ahead of the device.
            launch.append(dict(group=gi, step=step, params=plist, table=table, n=n,
                               hyper=torch.zeros(8, dtype=torch.float32, device=dev),
                               ring=[torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(4)],
                               events=[None] * 4, slot=0))
        self._launch = launch

    def load_state_dict(self, state_dict):
        """torch's loader replaces `state[p]["exp_avg"/"exp_avg_sq"]` with fresh tensors: drop the arenas so the next step rebuilds
        them and re-adopts the LOADED moments (otherwise the kernel would keep updating the old arena while state_dict()
        serialised the stale loaded tensors)."""
        super().load_state_dict(state_dict)
        # parameters / gradients keep viewing the old arenas (still valid memory) until the rebuild copies them over
        if self._arena is not None:
            self._written.clear()
        self._arena = None
        self._launch = None

    # ---- the step -------------------------------------------------------------------------------------------------
    def _upload_hyper(self):
        """Host -> device upload of (beta1, beta2, eps, step_size, lr) for every launch class and advance of the step
        counters.  Eager by design: under hipGraph replay (`capturing=True` steps) this is the only per-step host work."""
        for c in self._launch:
            g = self.param_groups[c["group"]]
            b1, b2 = g["betas"]
            t = c["step"] + 1
            step_size = g["lr"]
            if g["correct_bias"]:
                step_size = step_size * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            k = c["slot"] = (c["slot"] + 1) % 4
            if c["events"][k] is not None:
                c["events"][k].synchronize()
            host = c["ring"][k]
            host[0], host[1], host[2], host[3], host[4] = b1, b2, g["eps"], step_size, g["lr"]
            c["hyper"].copy_(host, non_blocking=True)
            c["events"][k] = torch.cuda.Event()
            c["events"][k].record()
            c["step"] = t
            for p in c["params"]:
                self.state[p]["step"] = t

    def _launch_kernels(self):
        a = self._arena
        for c in self._launch:
            ops.adamw_step(a["p"], a["g"], a["m"], a["v"], c["table"], c["n"], c["hyper"], self.grad_scale, p_bf16=a["pb"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture of a whole training step: only the device work is recorded; the caller uploads the
            # hyper-parameters eagerly before every replay (`prepare_replay()`).  The arena must already exist.
            if self._arena is None or self._launch is None:
                raise RuntimeError("run at least one eager optimizer step before capturing a step into a graph")
            if self._arena["ids"] != [id(p) for _, p in self._members()]:
                raise RuntimeError("the set of parameters receiving gradients changed; cannot capture")
            self._ensure_arena()          # records the device copies that adopt the few non-arena gradients (embeddings)
            self._written.clear()
            self._launch_kernels()
            return loss
        members = self._ensure_arena()
        if not members:
            return loss
        if self._launch is None or any(self.state[c["params"][0]]["step"] != c["step"] for c in self._launch):
            self._build_launch(members)
        a = self._arena
        self._written.clear()                                   # gradients are consumed: slots may be written directly again
        if getattr(self, "grad_sync", None) is not None:      # data-parallel gradient exchange (ytvln/distributed.py)
            self.grad_sync(a["g"], [(p,) + a["index"][id(p)] for _, p in members])
        self._upload_hyper()
        self._launch_kernels()
        return loss

    # -- two-phase capture for data-parallel runs (ytvln.distributed.GraphedTrainStep): the gradient exchange sits between the
    #    "gradients are complete in the arena" point and the update, and stays outside any graph.
    def capture_adopt(self):
        """Inside a capture, after backward: record the copies that bring the few non-arena gradients into the flat arena."""
        if not torch.cuda.is_current_stream_capturing():
            raise RuntimeError("capture_adopt() is only meaningful while capturing a hipGraph")
        if self._arena is None or self._launch is None:
            raise RuntimeError("run at least one eager optimizer step before capturing a step into a graph")
        if self._arena["ids"] != [id(p) for _, p in self._members()]:
            raise RuntimeError("the set of parameters receiving gradients changed; cannot capture")
        self._ensure_arena()

    def capture_adopt_some(self, params):
        """Inside a capture, in the middle of a phased backward: bring the gradients of `params` (complete at this point, the others may
        not exist yet) into their arena slots -- the partial form of capture_adopt()."""
        if not torch.cuda.is_current_stream_capturing():
            raise RuntimeError("capture_adopt_some() is only meaningful while capturing a hipGraph")
        if self._arena is None or self._launch is None:
            raise RuntimeError("run at least one eager optimizer step before capturing a step into a graph")
        fg, index = self._arena["g"], self._arena["index"]
        base_g = fg.data_ptr()
        with torch.no_grad():
            for p in params:
                if p.grad is None or id(p) not in index:
                    continue
                o, n = index[id(p)]
                if p.grad.data_ptr() != base_g + 4 * o:
                    fg[o:o + n].copy_(p.grad.reshape(-1))
                    p.grad = fg[o:o + n].view(p.shape)

    # -- the update in pieces (ytvln.distributed.GraphedTrainStep, phased): each group of arena ranges is updated as soon as ITS gradients are
    #    complete and exchanged, on the communication stream, under the rest of the backward pass.  The kernel is element-wise over a chunk
    #    table, so cutting the table changes no bit.
    def group_tables(self, group_slices, owner=None):
        """[per group: [(launch class index, chunk table on the device, number of chunks)]] for groups given as lists of (lo, hi) ranges of
        the flat arena (whole parameter slots).  Cached on `owner` until the launch classes are rebuilt."""
        if self._arena is None or self._launch is None:
            raise RuntimeError("run at least one eager optimizer step first")
        cache = getattr(owner, "_group_tables_cache", None) if owner is not None else None
        if cache is not None and cache[0] is self._launch:
            return cache[1]
        index, dev = self._arena["index"], self._arena["p"].device
        starts = sorted((lo, hi, k) for k, sl in enumerate(group_slices) for lo, hi in sl)

        def group_of(o):
            for lo, hi, k in starts:
                if lo <= o < hi:
                    return k
            return None
        out = [[] for _ in group_slices]
        covered = 0
        for ci, c in enumerate(self._launch):
            wd = float(self.param_groups[c["group"]]["weight_decay"])
            recs = [bytearray() for _ in group_slices]
            counts = [0] * len(group_slices)
            for p in c["params"]:
                o, numel = index[id(p)]
                k = group_of(o)
                if k is None:
                    raise RuntimeError("a parameter of the arena belongs to no gradient group: cannot split the update")
                covered += numel
                for ch in range(0, numel, CHUNK):
                    recs[k] += struct.pack("<qqff", o + ch, min(CHUNK, numel - ch), wd, 0.0)
                    counts[k] += 1
            for k, rec in enumerate(recs):
                if counts[k]:
                    out[k].append((ci, torch.frombuffer(rec, dtype=torch.uint8).to(dev), counts[k]))
        if owner is not None:
            owner._group_tables_cache = (self._launch, out)
        return out

    def launch_tables(self, tables):
        """The fused AdamW kernels of one group, on the current stream (hyper-parameters come from prepare_replay())."""
        a = self._arena
        for ci, table, n in tables:
            ops.adamw_step(a["p"], a["g"], a["m"], a["v"], table, n, self._launch[ci]["hyper"], self.grad_scale, p_bf16=a["pb"])

    def finish_group_step(self):
        """After the last group of a step: gradients are consumed, arena slots may be written directly again."""
        self._written.clear()

    def arena_range(self, p):
        """(offset, numel) of a parameter's slot in the flat arenas, or None."""
        return None if self._arena is None else self._arena["index"].get(id(p))

    def capture_update(self):
        """Inside a capture: record the fused AdamW kernels (hyper-parameters come from prepare_replay())."""
        if not torch.cuda.is_current_stream_capturing():
            raise RuntimeError("capture_update() is only meaningful while capturing a hipGraph")
        self._written.clear()
        self._launch_kernels()

    def flat_grad(self):
        """The flat fp32 gradient arena (None before the first optimizer step)."""
        return None if self._arena is None else self._arena["g"]

    def prepare_replay(self):
        """Call before each replay of a captured training step (after scheduler.step() set the new learning rate)."""
        self._upload_hyper()

    def zero_grad(self, set_to_none: bool = True):
        """Drop the gradients (`p.grad = None`).  With the arena in place the next backward writes weight gradients straight
        into their arena slots (ytvln.ops._direct_grad) and autograd adopts those views, so no memset and no accumulation
        pass is needed; the few small tensors that arrive as separate allocations (biases, LayerNorm, embeddings) are
        copied into their slots at the next step().  `set_to_none=False` zeroes the arena in place and keeps the views."""
        if self._arena is not None:
            self._written.clear()
        if self._arena is None or set_to_none:
            return super().zero_grad(set_to_none=True)
        self._arena["g"].zero_()
