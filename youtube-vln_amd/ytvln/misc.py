"""The two seeding / rank helpers of the reference's `utils/misc.py` that the training scripts call around the hot path
(pretrain.py:38, train.py:38).  Logging, output-directory naming and experiment bookkeeping stay with the reference."""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops


def set_seed(args) -> None:
    """utils/misc.py:36-44: `args.seed` (+ local rank when data parallel, so the ranks draw different masks) seeds torch, numpy and
    `random`.  The dropout / on-device masking stream of this package restarts from the same value (it would otherwise keep the state it
    derived from torch's initial seed when it was first used)."""
    if not getattr(args, "seed", None):
        return
    seed = int(args.seed)
    if getattr(args, "local_rank", -1) != -1:
        seed += int(args.local_rank)
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    ops.DropoutState.manual_seed(None)          # follow torch.initial_seed() (= seed) from here on, counters restarted


def is_default_gpu(args) -> bool:
    """utils/misc.py:46-50, as written there: true without data parallelism, and on every rank but 0 with it."""
    return getattr(args, "local_rank", -1) == -1 or dist.get_rank() != 0


def set_host_wait(blocking: bool = True, device=None) -> None:
    """How this process waits for its GPU.  One process per GPU (utils/distributed.py:63-104) means N ranks share one host's cores: with the
    runtime's default a rank that has filled its stream's queue, or sits in `torch.cuda.synchronize()`, SPINS on a core (measured round 5:
    221 ms of process CPU per 110 ms step = two cores per rank).  `blocking=True` = hipDeviceScheduleBlockingSync through the C ABI
    (`ytvln_set_host_wait`): waiting threads sleep on an interrupt.  Host-side only; call it after `torch.cuda.set_device(...)` and before
    the training loop.  Pair it with `StepPacer` so that the enqueueing thread waits on a blocking EVENT instead of a full queue."""
    if device is None:
        device = torch.cuda.current_device()
    device = torch.device("cuda", device).index if not isinstance(device, int) else device
    _lib.call("ytvln_set_host_wait", int(device), 1 if blocking else 0)


class StepPacer:
    """Keeps the host at most `depth` steps ahead of the device and lets it SLEEP while it waits: after step i has been enqueued the host waits
    for the end of step i - depth.  mode "event": hipEventSynchronize on a blocking event (sleeps on an interrupt under
    `set_host_wait(True)`); mode "poll": `event.query()` every `poll_ms` with `time.sleep` in between (never spins, whatever the runtime's wait
    policy; costs up to poll_ms of latency, which `depth` >= 1 hides).  `tick()` after every enqueued step, `drain()` at the end."""

    def __init__(self, depth: int = 2, mode: str = "event", poll_ms: float = 1.0):
        import collections
        if mode not in ("event", "poll"):
            raise ValueError(mode)
        self.depth = max(0, int(depth))
        self.mode, self.poll_s = mode, poll_ms * 1e-3
        self._events = collections.deque()

    def _wait(self, ev) -> None:
        if self.mode == "event":
            ev.synchronize()
            return
        import time
        while not ev.query():
            time.sleep(self.poll_s)

    def tick(self, stream=None) -> None:
        ev = torch.cuda.Event(blocking=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self._events.append(ev)
        while len(self._events) > self.depth:
            self._wait(self._events.popleft())

    def drain(self) -> None:
        while self._events:
            self._wait(self._events.popleft())
