"""Generate tests/golden/g14_trainmode_stats.npz and g15_tiny_traj20.npz by running the REAL reference (imported from /root/reference).

TEST INFRASTRUCTURE ONLY; runs in the build container only.  Usage:  python oracle/gen_golden_trainmode.py [g14 g15]

g14 -- train-mode (dropout ON) agreement "in distribution" (SURVEY.md H1).  The reference's dropout masks come from torch's global
       Philox stream (`vilbert/vilbert.py:238,274,319,362,403,447,...`), the HIP path draws its own stream, so a bit-level comparison
       does not exist.  What is comparable is the DISTRIBUTION of the loss trajectory over mask seeds: the tiny config with the config's
       own p = 0.1 everywhere, one fixed batch, fixed initial weights, the reference's `get_optimization` AdamW + WarmupLinear
       (`utils/utils_init.py:199-239` loop body), 20 optimizer steps, N_SEEDS mask seeds (`torch.manual_seed(seed)` before the model is
       put in train mode).  Stored: every loss of every seed and step [seeds, steps, 5] (total, ranking, traj, vision, language) -- the GPU
       test compares the mean over ITS seeds with the mean over these, in units of the reference's own standard deviation.
g15 -- the same recipe with p = 0 (no `opt_mask` hole, all four heads): a FINITE 20-step loss trajectory + per-tensor norms / sums of the
       parameters after step 3 and step 20 (g0's multi-step trajectory is NaN in the reference itself because of its `opt_mask` hole).
The oracle restatement runs beside the reference for g15 and the script aborts on disagreement.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (shares the recipes; importing it does not run anything)
import vilbert_ref as O  # noqa: E402
from ytvln import synth  # noqa: E402

TASKS = ("ranking", "traj", "vision", "language")
RECIPE = dict(bs=2, K=3, T=16, frames=2, boxes=4, seed=31, ignore_rank_frac=0.0)
STEPS, LR, TOTAL_STEPS, W_SEED = 20, 1e-3, 40, 16
N_SEEDS = 128


def _run(R, rcfg, args, W, batch, seed, steps, dropout_prob):
    rcfg.args = args
    torch.manual_seed(seed)
    model = R.lily.Lily(rcfg, dropout_prob=dropout_prob)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    model.train()
    a = G.ref_args(**{**vars(args), "learning_rate": LR})
    opt, sched, _, _ = R.vilbert_init.get_optimization(a, model, TOTAL_STEPS, None)
    torch.manual_seed(seed)            # the mask stream starts here (weights are loaded, not drawn)
    rows = []
    for _ in range(steps):
        outputs = model(*R.utils_init.get_model_input(batch))
        total, per = G.ref_losses(R, batch, outputs, a)
        total.backward()
        rows.append([float(total)] + [float(per[t]) for t in TASKS])
        opt.step(); sched.step(); model.zero_grad()
    return np.array(rows, np.float64), model, sched


def g14(R):
    rcfg, _ = G.load_cfg(R, "tiny_2_2_1.json")
    args = G.ref_args(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    rcfg.args = args
    shapes = {k: tuple(v.shape) for k, v in R.lily.Lily(rcfg).state_dict().items()}
    W = synth.make_weights(shapes, W_SEED)
    batch = synth.to_torch(synth.make_batch(**RECIPE))
    t0 = time.time()
    allrows = np.stack([_run(R, rcfg, args, W, batch, s, STEPS, 0.1)[0] for s in range(N_SEEDS)])
    assert np.isfinite(allrows).all()
    out = dict(losses=allrows.astype(np.float32), mean=allrows.mean(0), std=allrows.std(0, ddof=1), n_seeds=np.int64(N_SEEDS),
               steps=np.int64(STEPS), lr=np.float64(LR), total_steps=np.int64(TOTAL_STEPS), w_seed=np.int64(W_SEED),
               names=np.array(("total",) + TASKS))
    np.savez_compressed(os.path.join(G.GOLD, "g14_trainmode_stats.npz"), **out)
    print(f"g14 ok ({time.time() - t0:.0f} s): total loss mean/std at steps 0, 9, 19:",
          [(round(float(out['mean'][i, 0]), 4), round(float(out['std'][i, 0]), 4)) for i in (0, 9, 19)])


def g15(R):
    rcfg, ocfg = G.load_cfg(R, "tiny_2_2_1.json", **G.ZERO_DROP)
    args = G.ref_args(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    rcfg.args = args
    shapes = {k: tuple(v.shape) for k, v in R.lily.Lily(rcfg, dropout_prob=0.0).state_dict().items()}
    W = synth.make_weights(shapes, W_SEED)
    batch = synth.to_torch(synth.make_batch(**RECIPE))
    out = {}
    rows3, model3, _ = _run(R, rcfg, args, W, batch, 0, 3, 0.0)
    out["post3_norm"] = np.array([p.double().norm().item() for _, p in model3.named_parameters()])
    out["post3_sum"] = np.array([p.double().sum().item() for _, p in model3.named_parameters()])
    rows, model, sched = _run(R, rcfg, args, W, batch, 0, STEPS, 0.0)
    assert np.isfinite(rows).all() and np.array_equal(rows[:3], rows3)
    # the oracle beside it
    S = G.state_of(W)
    ost = O.AdamWState()
    warm, tot = O.schedule_totals(TOTAL_STEPS, 1, 1)
    flags = G.flags_of(args)
    for step in range(STEPS):
        oloss, _, _, _ = O.train_step(S, ocfg, flags, batch, ost, LR * O.warmup_linear(step, warm, tot))
        G.check(f"g15 step {step}", rows[step, 0], oloss, 2e-5 * (step + 1), 0)
    out["losses"] = rows
    out["names"] = np.array(("total",) + TASKS)
    out["post_names"] = np.array([n for n, _ in model.named_parameters()])
    out["post_norm"] = np.array([p.double().norm().item() for _, p in model.named_parameters()])
    out["post_sum"] = np.array([p.double().sum().item() for _, p in model.named_parameters()])
    out["steps"], out["lr"], out["total_steps"], out["w_seed"] = np.int64(STEPS), np.float64(LR), np.int64(TOTAL_STEPS), np.int64(W_SEED)
    np.savez_compressed(os.path.join(G.GOLD, "g15_tiny_traj20.npz"), **out)
    print("g15 ok: total loss", [round(float(x), 4) for x in rows[::4, 0]])


if __name__ == "__main__":
    import ref_import
    R = ref_import.import_reference()
    for name in (sys.argv[1:] or ["g15", "g14"]):
        globals()[name](R)
