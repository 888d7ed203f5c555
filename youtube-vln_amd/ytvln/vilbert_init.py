"""Optimizer / scheduler factory: the reference's parameter grouping, schedule arithmetic and resume semantics
(`vilbert/vilbert_init.py:7-72`), split into the three things it does."""
from __future__ import annotations

from pathlib import Path

import torch
from torch import nn

from . import ops
from .optimization import AdamW, ConstantLRSchedule, WarmupLinearSchedule

# substring match, exactly as the reference: "biOutput.LayerNorm1.weight" / "LayerNorm2.weight" do NOT match and are
# therefore weight-decayed (vilbert_init.py:9-18).
NO_DECAY = ("bias", "LayerNorm.weight", "LayerNorm.bias")


def grouped_parameters(model: nn.Module, weight_decay: float):
    """[{no-decay tensors, wd 0}, {the rest, wd}] in named_parameters() order (vilbert_init.py:9-18)."""
    plain, decayed = [], []
    for name, param in model.named_parameters():
        (plain if any(tag in name for tag in NO_DECAY) else decayed).append(param)
    return [{"params": plain, "weight_decay": 0.0}, {"params": decayed, "weight_decay": weight_decay}]


def build_scheduler(args, optimizer, steps_per_epoch: int):
    """Constant LR for `--ConstantLR` / pre-training `--no_scheduler`; otherwise linear warm-up over `warmup_proportion` of the
    optimizer steps and a linear decay stretched by `cooldown_factor` (vilbert_init.py:23-40).  The reference's fine-tune
    `--no_scheduler` branch names an un-imported MultiplicativeLR(λ = 1) and cannot run; its meaning is a constant schedule."""
    if args.ConstantLR or args.no_scheduler:
        return ConstantLRSchedule(optimizer)
    optimizer_steps = (steps_per_epoch // args.gradient_accumulation_steps) * args.num_epochs
    warmup = args.warmup_proportion * optimizer_steps
    horizon = warmup + args.cooldown_factor * (optimizer_steps - warmup)
    return WarmupLinearSchedule(optimizer, warmup_steps=warmup, t_total=horizon, last_epoch=-1)


def restore_checkpoint(path, model, optimizer, scheduler, logger=None) -> int:
    """Load whatever of {model_state_dict, optimizer_state_dict, scheduler_state_dict, epoch} the file holds
    (vilbert_init.py:44-64); returns the epoch to continue from.  A missing file is reported, not fatal, like the reference.
    Files written by this repo also carry the dropout / masking stream position (`ytvln_rng_state`)."""
    path = Path(path)
    say = logger.info if logger else (lambda *_: None)
    say(f"resume the training model from {path}")
    if not path.exists():
        say(f"resuming the training model failed, {path} does not exist")
        return 0
    ckpt = torch.load(path, map_location="cpu")
    net = model.module if isinstance(getattr(model, "module", None), nn.Module) else model
    loaders = (("model_state_dict", net.load_state_dict), ("optimizer_state_dict", optimizer.load_state_dict),
               ("scheduler_state_dict", scheduler.load_state_dict))
    for key, load in loaders:
        if key in ckpt:
            load(ckpt[key])
            say(f"load {key}...")
    if "ytvln_rng_state" in ckpt:
        p0 = next(net.parameters(), None)          # the mask stream lives on the model's device, which need not be the current one
        ops.DropoutState.set_state(ckpt["ytvln_rng_state"], device=p0.device if (p0 is not None and p0.is_cuda) else None)
    return ckpt["epoch"] + 1 if "epoch" in ckpt else 0


def get_optimization(args, model, train_data_loader_length, logger):
    """-> (optimizer, scheduler, model, start_epoch), the reference's signature and return order."""
    optimizer = AdamW(grouped_parameters(model, args.weight_decay), lr=args.learning_rate)
    scheduler = build_scheduler(args, optimizer, train_data_loader_length)
    start_epoch = 0
    if getattr(args, "resume", False):
        start_epoch = restore_checkpoint(args.from_pretrained, model, optimizer, scheduler, logger)
        if args.ConstantLR:                   # keep the learning rate the loaded run ended on (vilbert_init.py:67-69)
            scheduler.base_lrs = scheduler._last_lr
    return optimizer, scheduler, model, start_epoch
