"""Batch parsing, the four losses and the training-step body with the reference's semantics (`utils/utils_init.py`).

`get_model_input` / `get_loss_correct` / `compute_metrics_independent` / `train_epoch` keep the reference names and
argument order.  Differences are internal only: the big reductions (30522-way CE, 1601-way masked KL) and the small ones
run on the fused loss kernels, and nothing calls `.item()` inside the step (the reference syncs at utils_init.py:127).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist

from . import ops


def pad_packed(t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """utils/dataset/common.py:21-26: scatter the packed [N] scores back to [bs, K], -inf where opt_mask is False."""
    mask = mask.bool()
    out = torch.full(mask.shape, -float("inf"), dtype=t.dtype, device=t.device)
    out[mask] = t
    return out


def val_args(args) -> None:
    """utils_init.py:13-23: at least one objective; when fine-tuning, the trajectory-judgement task and `--shuffle_visual_features`
    come together unless ranking / `--not_traj_judge_data` supplies the data."""
    if not (args.masked_vision or args.masked_language or args.ranking or args.traj_judge):
        raise ValueError("No training objective selected, add --masked_vision, --masked_language, --ranking, or --traj_judge")
    if not args.pretrain and args.traj_judge and (bool(args.ranking or args.not_traj_judge_data) != bool(args.shuffle_visual_features)):
        raise ValueError("fine-tuning with --traj_judge: drop --shuffle_visual_features, or run both tasks and pass it")


def get_time() -> str:
    """utils_init.py:26-27."""
    from datetime import datetime
    return datetime.now().strftime("%Y-%m-%d %H:%M")


def get_model_input(batch, all_options=None):
    """utils_init.py:34-77: unpack the 16-tuple, drop padded options with opt_mask, return Lily.forward's 9 positionals.

    `all_options=True` (known on the host before the H2D copy, see train_epoch) means opt_mask has no holes: the
    [bs, K, ...] tensors are then flattened as views instead of the reference's boolean-mask gather, which costs a
    device->host sync (nonzero) plus a second copy of the region features."""
    (_, image_features, image_locations, image_mask, _, _, instr_tokens, instr_mask, _, instr_highlights, segment_ids,
     co_attention_mask, _, opt_mask, _, attend_order_visual_feature) = batch
    if all_options is None:
        all_options = (not opt_mask.is_cuda) and bool(opt_mask.all())
    if all_options:
        flat = lambda x: x.flatten(0, 1)      # noqa: E731
    else:
        flat = lambda x: x[opt_mask]          # noqa: E731
    co_attention_mask = co_attention_mask.view(-1, co_attention_mask.size(2), co_attention_mask.size(3))
    return (flat(instr_tokens), flat(image_features), flat(image_locations), flat(segment_ids), flat(instr_mask),
            flat(image_mask), co_attention_mask, flat(instr_highlights), attend_order_visual_feature)


def get_device(batch):
    return batch[0].device


def get_mask_options(batch) -> torch.Tensor:
    return batch[13]


def get_batch_size(batch):
    return batch[1].size(0)


def get_ranking_target(batch):
    return batch[0]


def get_vision_target(batch, all_options=False):
    if all_options:
        return batch[4].flatten(0, 2), batch[5].flatten()
    opt_mask = get_mask_options(batch)
    return batch[4][opt_mask].flatten(0, 1), batch[5][opt_mask].flatten()


def get_linguistic_target(batch, all_options=False):
    if all_options:
        return batch[8].flatten()
    return batch[8][get_mask_options(batch)].flatten()


def get_loss_correct(batch: List[torch.Tensor], outputs: Dict[str, torch.Tensor], task, args, logger, training,
                     all_options=False):
    """utils_init.py:108-164 -> (batch_size, target, loss, correct).  `all_options`: see get_model_input."""
    opt_mask = get_mask_options(batch)
    unpack = (lambda t: t.view(opt_mask.shape)) if all_options else (lambda t: pad_packed(t, opt_mask))
    batch_size = get_batch_size(batch)
    device = opt_mask.device
    correct = torch.zeros((), device=device)        # (device-side fill: safe inside hipGraph capture, unlike a host scalar copy)
    if task == "vision":
        predictions = outputs["vision"]
        predictions = predictions.reshape(-1, predictions.shape[2])
        target, target_mask = get_vision_target(batch, all_options)
        loss = ops.kl_masked(predictions, target.float(), target_mask)        # sum(KL*mask) / max(1, sum(mask)), on device
    elif task == "language":
        voc_size = outputs["language"].shape[-1]
        target = get_linguistic_target(batch, all_options)
        loss = ops.cross_entropy(outputs["language"].reshape(-1, voc_size), target, ignore_index=-1)
    elif task == "ranking":
        target = get_ranking_target(batch)
        prediction = unpack(outputs["ranking"].squeeze(1))
        if training:
            loss = ops.cross_entropy(prediction, target, ignore_index=-1)
            correct = torch.sum(torch.argmax(prediction, 1) == target).float()
        else:
            loss = ops.bce_with_logits(prediction, target.float())
            correct = torch.sum(target.gather(1, torch.argmax(prediction, 1).view(-1, 1))).float()
    elif task == "traj":
        prediction = unpack(outputs["traj"].squeeze(1))
        target = torch.zeros(prediction.shape, device=device).bool()
        if not (args.ranking or args.not_traj_judge_data):
            target[:, 0] = 1
        elif args.pretrain:
            target[:, :(1 + args.num_negatives)] = 1
        else:
            target[:, :-args.num_negatives] = 1
        pos_weight = (target.shape[1] / target[0].sum() - 1).reshape(1).float()      # negatives / positives, stays on device
        loss = ops.bce_with_logits(prediction, target.float(), pos_weight)
        correct = torch.sum((prediction.sigmoid() > 0.5) == target).float() / target.shape[1]
    else:
        raise KeyError(task)
    return batch_size, target, loss, correct


def compute_metrics_independent(batch, outputs, task, args, logger, reduced_metrics, all_options=False) -> torch.Tensor:
    """utils_init.py:167-189: loss for backward + (optionally all-reduced) logging copies."""
    device = get_device(batch)
    batch_size, target, loss, correct = get_loss_correct(batch, outputs, task, args, logger, True, all_options)
    reduced_loss = loss.detach().float()
    reduced_correct = correct.detach().float()
    reduced_batch_size = torch.full((), float(batch_size), device=device)
    if getattr(args, "local_rank", -1) != -1 and not getattr(args, "skip_all_reduce", False) and dist.is_initialized():
        from . import distributed as D
        packed = torch.stack([reduced_loss / float(D.metrics_world_size()), reduced_correct, reduced_batch_size])
        D.metrics_all_reduce_(packed)          # one small collective instead of the reference's three; on the RCCL data plane when there is one
        reduced_loss, reduced_correct, reduced_batch_size = packed[0], packed[1], packed[2]
    reduced_metrics["loss"][task] = reduced_loss
    if task not in ("vision", "language"):
        reduced_metrics["accuracy"][task] = reduced_correct / reduced_batch_size
    return loss


TASKS = (("vision", "masked_vision"), ("language", "masked_language"), ("ranking", "ranking"), ("traj", "traj_judge"))


def _head_capacity(rows: int, frac: float) -> int:
    """Static number of rows sent through a prediction head: frac * rows rounded up to a whole 128-row GEMM tile."""
    return min(rows, max(1, -(-int(rows * frac) // 128)) * 128)


def _loss_aware_step(model, batch, args, all_options, capacity_frac):
    """Forward + losses with the prediction heads evaluated only on rows that carry a target.  Identical
    losses and gradients: ignored tokens / unmasked regions contribute nothing to CE / KL.  Row selection has a static shape
    (`capacity_frac` of the rows; Bernoulli(0.15) masking makes 0.25 a > 20 sigma bound) and needs no host sync."""
    inputs = get_model_input(batch, all_options)
    rows, lm_t, vis_t, vis_m = {}, None, None, None
    if args.masked_language:
        lm_t = get_linguistic_target(batch, bool(all_options))
        cap = _head_capacity(lm_t.numel(), capacity_frac)
        rows["language"] = ops.select_rows(lm_t != -1, cap)
    if args.masked_vision:
        vis_t, vis_m = get_vision_target(batch, bool(all_options))
        cap = _head_capacity(vis_m.numel(), capacity_frac)
        rows["vision"] = ops.select_rows(vis_m == 1, cap)
    outputs = model(*inputs, head_rows=rows)
    losses = {}
    if args.masked_vision:
        idx = rows["vision"]
        losses["vision"] = ops.kl_masked(outputs["vision"], ops.gather_rows(vis_t.float(), idx), vis_m[idx])
    if args.masked_language:
        losses["language"] = ops.cross_entropy(outputs["language"], lm_t[rows["language"]], ignore_index=-1)
    overflow = sum(((t != ign).sum() > r.numel()).float() for t, ign, r in
                   ((lm_t, -1, rows.get("language")), (vis_m, 0, rows.get("vision"))) if t is not None)
    return outputs, losses, overflow


def train_step(model, optimizer, scheduler, batch, args, step: int = 0, logger=None, all_options=None,
               loss_aware_heads: bool = False, capacity_frac: float = 0.25, optimizer_step: bool = True, backward=None):
    """One iteration of train_epoch's body (utils_init.py:199-239): forward, loss composition in the reference's order,
    backward, and -- every gradient_accumulation_steps -- optimizer.step(); scheduler.step(); zero_grad().
    Returns (loss, reduced_metrics) as device tensors; never synchronises the host.
    `backward` (optional callable taking the loss) replaces `loss.backward()` -- ytvln.distributed.GraphedTrainStep runs the backward
    pass in phases through it."""
    reduced_metrics = {"loss": {}, "accuracy": {}}
    fused = {}
    if loss_aware_heads and (args.masked_language or args.masked_vision):
        outputs, fused, overflow = _loss_aware_step(model, batch, args, all_options, capacity_frac)
        reduced_metrics["head_row_overflow"] = overflow        # device scalar; > 0 means capacity_frac was too small
    else:
        outputs = model(*get_model_input(batch, all_options))
    loss = None
    for task, flag in TASKS:
        if getattr(args, flag):
            if task in fused:
                l = fused[task]
                reduced_metrics["loss"][task] = l.detach()
            else:
                l = compute_metrics_independent(batch, outputs, task, args, logger, reduced_metrics, bool(all_options))
            if task == "traj":
                l = args.traj_loss_scale * l
            loss = l if loss is None else loss + l
    accum = getattr(args, "gradient_accumulation_steps", 1)
    if accum > 1:
        loss = loss / accum
        if hasattr(model, "require_backward_grad_sync"):
            # data parallel: exchange the ACCUMULATED gradients once, during the last micro-step's backward (a bucket reduced after
            # the first micro-step would be summed over ranks and then have un-reduced local gradients added to it)
            model.require_backward_grad_sync = (step + 1) % accum == 0
    if backward is not None:
        backward(loss)
    else:
        loss.backward()
    ops.TwoStream.join_backward()       # two-stream mode: the text side's backward kernels before the optimizer reads the gradients
    if not optimizer_step:          # forward/backward only (ytvln.distributed.GraphedTrainStep captures the update separately)
        return loss.detach(), reduced_metrics
    if (step + 1) % accum == 0:
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        if hasattr(optimizer, "_arena"):
            optimizer.zero_grad()              # keeps the flat-arena views (one memset)
        else:
            model.zero_grad()
    return loss.detach(), reduced_metrics


def train_epoch(epoch, model, optimizer, scheduler, data_loader, writer, default_gpu, args, logger) -> None:
    """utils_init.py:192-268.  Batches are moved to the model's device; scalars are only formatted (host sync) on the
    logging rank, once per step, after the step has been queued."""
    device = next(model.parameters()).device
    model.train()
    model.zero_grad()
    from .misc import StepPacer
    pacer = StepPacer(2)          # the host sleeps on a blocking event two steps back instead of spinning for queue room (one core per rank)
    for step, batch in enumerate(data_loader):
        all_options = bool(batch[13].all()) if not batch[13].is_cuda else None      # decided on the host, no GPU sync
        batch = tuple(t.to(device, non_blocking=True) if hasattr(t, "to") else t for t in batch)
        loss, reduced_metrics = train_step(model, optimizer, scheduler, batch, args, step, logger, all_options)
        pacer.tick()
        if default_gpu and writer is not None:
            global_step = step + epoch * len(data_loader)
            if "head_row_overflow" in reduced_metrics and float(reduced_metrics["head_row_overflow"]) > 0:
                # loss-aware heads (opt-in) decode a fixed number of rows; more target-carrying rows than that would be dropped from
                # the loss -- never silently (checked here, where the step's scalars are read back anyway)
                raise RuntimeError("loss_aware_heads: more rows carry a target than the row capacity; raise capacity_frac or use the full heads")
            total = sum(reduced_metrics["loss"].values())
            writer.add_scalar("learning_rate/train", float(scheduler.get_last_lr()[0]), global_step=global_step)
            writer.add_scalar("loss/train", float(total), global_step=global_step)
            for task, item in reduced_metrics["accuracy"].items():
                writer.add_scalar(f"accuracy/{task}", float(item), global_step=global_step)
            for task, item in reduced_metrics["loss"].items():
                writer.add_scalar(f"loss/{task}", float(item), global_step=global_step)


# ------------------------------------------------------------------------------------------------------------------
# checkpoints and evaluation loops (reference utils/utils_init.py:273-446)
# ------------------------------------------------------------------------------------------------------------------
import os

from torch import nn


def get_model_path(model_save_path, save_name):
    return os.path.join(model_save_path, f"{save_name}.bin")


def save_model(model_save_path, save_name, logger, model, optimizer, scheduler, epoch):
    """utils_init.py:277-295: `{model_state_dict, optimizer_state_dict, scheduler_state_dict, epoch}` in one `.bin`.
    The state-dict keys and the per-parameter optimizer state (`step`, `exp_avg`, `exp_avg_sq`) are the reference's, so the
    file resumes in either implementation (tensors are cloned out of the flat arenas so the file stays self-contained)."""
    net = model.module if hasattr(model, "module") and isinstance(model.module, nn.Module) else model
    if not isinstance(net, nn.Module):
        raise ValueError("Can't find the Module here")
    if logger:
        logger.info(f"saving the {save_name} model")
    # state_dict() hands out the optimizer's LIVE per-parameter dicts (arena views): build a detached copy, never assign into them
    live = optimizer.state_dict()
    opt_state = {"param_groups": live["param_groups"],
                 "state": {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for i, st in live["state"].items()}}
    # the four keys are the reference's; `ytvln_rng_state` (dropout / masking stream position) is an extra the reference ignores
    torch.save({"model_state_dict": {k: v.detach().clone() for k, v in net.state_dict().items()},
                "optimizer_state_dict": opt_state, "scheduler_state_dict": scheduler.state_dict(), "epoch": epoch,
                "ytvln_rng_state": ops.DropoutState.get_state()},
               get_model_path(model_save_path, save_name))


def delete_model(model_save_path, save_name):
    model_path = get_model_path(model_save_path, save_name)
    if os.path.exists(model_path):
        os.unlink(model_path)


def _to_device(batch, device):
    return tuple(t.to(device, non_blocking=True) if hasattr(t, "to") else t for t in batch)


def val_independent(batch, outputs, task, args, logger, stats) -> None:
    """utils_init.py:306-312: add one batch to stats[task] = [rows, summed loss, correct, batches] (device tensors, no host sync)."""
    batch_size, _, loss, correct = get_loss_correct(batch, outputs, task, args, logger, False)
    stats[task] += torch.stack([torch.full((), float(batch_size), device=loss.device), loss.float(), correct.float().reshape(()),
                                torch.ones((), device=loss.device)])


def test_epoch(epoch: int, model, tag, data_loader, writer, default_gpu, args, global_step, logger):
    """utils_init.py:315-379: eval-mode ranking / traj losses and success rates, accumulated on the device."""
    device = next(model.parameters()).device
    model.eval()
    stats = {}
    if args.ranking:
        stats["ranking"] = torch.zeros(4, device=device)
    if args.traj_judge:
        stats["traj"] = torch.zeros(4, device=device)
    with torch.no_grad():
        for batch in data_loader:
            all_options = bool(batch[13].all()) if not batch[13].is_cuda else None
            batch = _to_device(batch, device)
            outputs = model(*get_model_input(batch, all_options))
            for task in stats:
                batch_size, _, loss, correct = get_loss_correct(batch, outputs, task, args, logger, False, bool(all_options))
                stats[task] += torch.stack([torch.full((), float(batch_size), device=device), loss.float(), correct.float(),
                                            torch.ones((), device=device)])
    reduced = {task: v.clone() for task, v in stats.items()}
    if getattr(args, "local_rank", -1) != -1 and not getattr(args, "skip_all_reduce", False) and dist.is_initialized():
        from . import distributed as D
        world = float(D.metrics_world_size())
        for task in reduced:
            D.metrics_all_reduce_(reduced[task])
            reduced[task][1] /= world
    for task in reduced:
        reduced[task][1] /= reduced[task][3]
        reduced[task][2] /= reduced[task][0]
    if default_gpu and writer is not None:
        for task, item in reduced.items():
            writer.add_scalar(f"loss/{task}_{tag}", float(item[1]), global_step=global_step)
            writer.add_scalar(f"accuracy/{task}_{tag}", float(item[2]), global_step=global_step)
    return reduced


def val_epoch(epoch: int, model, tag, data_loader, writer, default_gpu, args, global_step, logger, task):
    """utils_init.py:382-446: beam re-ranking validation -- BCE-with-logits against the multi-hot target and the success
    rate of the top-scoring beam."""
    device = next(model.parameters()).device
    model.eval()
    stats = torch.zeros(3, device=device)
    steps = 0
    with torch.no_grad():
        for batch in data_loader:
            all_options = bool(batch[13].all()) if not batch[13].is_cuda else None
            batch = _to_device(batch, device)
            outputs = model(*get_model_input(batch, all_options))
            opt_mask = get_mask_options(batch)
            target = get_ranking_target(batch)
            logit = outputs[task].squeeze(1).view(opt_mask.shape) if all_options else pad_packed(outputs[task].squeeze(1), opt_mask)
            loss = ops.bce_with_logits(logit, target.float())
            correct = torch.sum(target.gather(1, torch.argmax(logit, 1).view(-1, 1))).float()
            stats += torch.stack([torch.full((), float(get_batch_size(batch)), device=device), loss.float(), correct])
            steps += 1
    if getattr(args, "local_rank", -1) != -1 and dist.is_initialized():
        from . import distributed as D
        D.metrics_all_reduce_(stats)
    success_rate = stats[2] / stats[0]
    if default_gpu and writer is not None:
        writer.add_scalar(f"loss/{task}_{tag}", float(stats[1] / max(steps, 1)), global_step=global_step)
        writer.add_scalar(f"accuracy/{task}_{tag}", float(success_rate), global_step=global_step)
    return success_rate


# ------------------------------------------------------------------------------------------------------------------
# inference re-ranking (test.py:144-192): forward only, eval mode, ranking head
# ------------------------------------------------------------------------------------------------------------------
def get_instr_ids(batch) -> List[str]:
    """test.py:200-202: "<path id>_<instruction index>" per dataset item."""
    return [f"{int(item[0])}_{int(item[1])}" for item in batch[12].tolist()]


def eval_epoch(model, data_loader, args, all_options=None):
    """test.py:144-166 -> [(instr_id, [score per candidate path])].  One device->host copy per batch (the scores)."""
    device = next(model.parameters()).device
    model.eval()
    all_scores = []
    with torch.no_grad():
        for batch in data_loader:
            instr_ids = get_instr_ids(batch)
            if getattr(args, "random_testing", False):
                vil_logit = torch.rand(batch[0].shape)
            else:
                batch = _to_device(batch, device)
                output = model(*get_model_input(batch, all_options))
                opt_mask = get_mask_options(batch)
                scores = output["ranking"].squeeze(1)
                vil_logit = scores.view(opt_mask.shape) if all_options else pad_packed(scores, opt_mask)
            for instr_id, logit in zip(instr_ids, vil_logit.tolist()):
                all_scores.append((instr_id, logit))
    return all_scores


def convert_scores(all_scores, beam_data, add_exploration_path: bool = False):
    """test.py:169-192: pick the best-scored beam per instruction.  `beam_data` = the parsed beam file (list of dicts with
    instr_id / ranked_paths / exploration_path)."""
    beams_of = {item["instr_id"]: item["ranked_paths"] for item in beam_data}
    explore = {item["instr_id"]: [[vp] for vp in item["exploration_path"]] for item in beam_data} if add_exploration_path else {}
    output = []
    for instr_id, scores in all_scores:
        idx = max(range(len(scores)), key=scores.__getitem__)
        beams = beams_of[instr_id]
        trajectory = list(explore.get(instr_id, []))
        if idx >= len(beams):        # a perturbation won: fake a wrong destination by stopping at the initial location
            trajectory = [beams[0][0]]
        else:
            trajectory += beams[idx]
        output.append({"instr_id": instr_id, "trajectory": trajectory})
    return output
