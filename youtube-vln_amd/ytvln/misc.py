"""The two seeding / rank helpers of the reference's `utils/misc.py` that the training scripts call around the hot path
(pretrain.py:38, train.py:38).  Logging, output-directory naming and experiment bookkeeping stay with the reference."""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.distributed as dist

from . import ops


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
