"""Optimizer / scheduler factory with the reference's grouping and schedule arithmetic (`vilbert/vilbert_init.py:7-72`)."""
from __future__ import annotations

from pathlib import Path

import torch
from torch import nn

from .optimization import AdamW, ConstantLRSchedule, WarmupLinearSchedule

# substring match, exactly as the reference: "biOutput.LayerNorm1.weight" / "LayerNorm2.weight" do NOT match and are
# therefore weight-decayed (vilbert_init.py:9-18).
NO_DECAY = ("bias", "LayerNorm.weight", "LayerNorm.bias")


def grouped_parameters(model: nn.Module, weight_decay: float):
    groups = [{"params": [], "weight_decay": 0.0}, {"params": [], "weight_decay": weight_decay}]
    for name, param in model.named_parameters():
        groups[0 if any(nd in name for nd in NO_DECAY) else 1]["params"].append(param)
    return groups


def get_optimization(args, model, train_data_loader_length, logger):
    optimizer = AdamW(grouped_parameters(model, args.weight_decay), lr=args.learning_rate)

    if (args.pretrain and args.no_scheduler) or args.ConstantLR:
        scheduler = ConstantLRSchedule(optimizer)
    else:
        t_total = (train_data_loader_length // args.gradient_accumulation_steps) * args.num_epochs
        warmup_steps = args.warmup_proportion * t_total
        adjusted_t_total = warmup_steps + args.cooldown_factor * (t_total - warmup_steps)
        # (the reference's `--no_scheduler` fine-tune branch references an un-imported MultiplicativeLR and cannot run,
        #  vilbert_init.py:39; the working equivalent is a constant schedule)
        scheduler = (WarmupLinearSchedule(optimizer, warmup_steps=warmup_steps, t_total=adjusted_t_total, last_epoch=-1)
                     if not args.no_scheduler else ConstantLRSchedule(optimizer))

    start_epoch = 0
    if getattr(args, "resume", False):
        checkpoint_path = Path(args.from_pretrained)
        if logger:
            logger.info(f"resume the training model from {checkpoint_path}")
        if checkpoint_path.exists():
            state_dict = torch.load(checkpoint_path, map_location="cpu")
            target = model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model
            if "model_state_dict" in state_dict:
                target.load_state_dict(state_dict["model_state_dict"])
            if "optimizer_state_dict" in state_dict:
                optimizer.load_state_dict(state_dict["optimizer_state_dict"])
            if "scheduler_state_dict" in state_dict:
                scheduler.load_state_dict(state_dict["scheduler_state_dict"])
            if "epoch" in state_dict:
                start_epoch = state_dict["epoch"] + 1
        elif logger:
            logger.info(f"resumimg the training model failed, {checkpoint_path} does not exist")
        if args.ConstantLR:
            scheduler.base_lrs = scheduler._last_lr
    return optimizer, scheduler, model, start_epoch
