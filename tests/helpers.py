import json
import os
import types

import numpy as np
import torch

from conftest import CFG_DIR, GOLD

ZERO_DROP = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0,
                 v_hidden_dropout_prob=0.0)


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def cfg_dict(name, **over):
    d = json.load(open(os.path.join(CFG_DIR, name)))
    d.update(over)
    return d


def args_ns(**kw):
    a = dict(model_name="vilbert", ranking=False, traj_judge=False, masked_vision=False, masked_language=False,
             pretrain=True, num_negatives=2, traj_loss_scale=1.0, not_traj_judge_data=False, local_rank=-1,
             skip_all_reduce=True, weight_decay=0.01, learning_rate=4e-5, no_scheduler=False, ConstantLR=False,
             gradient_accumulation_steps=1, num_epochs=1, warmup_proportion=0.2, cooldown_factor=2.0, resume=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def close(a, b, atol, rtol, what=""):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = ~((err <= tol) | (torch.isnan(a) & torch.isnan(b)) | ((a == b)))
    assert not bool(bad.any()), f"{what}: max abs err {float(err[~torch.isnan(err)].max() if (~torch.isnan(err)).any() else 0):.3e}, " \
                                f"{int(bad.sum())}/{a.numel()} outside atol={atol} rtol={rtol}"
    return float(err[~torch.isnan(err)].max()) if (~torch.isnan(err)).any() else 0.0


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))
