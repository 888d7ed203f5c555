"""Checkpoint interoperability with the REAL reference, both directions.  TEST INFRASTRUCTURE ONLY; build container only.

    python oracle/gen_golden_ckpt.py write     # reference -> fixture
    python oracle/gen_golden_ckpt.py verify    # repo-written checkpoint (made on the GPU box by tools/write_repo_ckpt.py) -> reference

write:  the reference's own `Lily` + `AdamW` + `WarmupLinearSchedule` (built by the reference's `get_optimization`) train the micro config
        for 2 steps; the checkpoint is written with the reference's `save_model` (utils/utils_init.py:277-295) to
        tests/golden/g9_ref_ckpt.bin; a THIRD reference step gives the expected continuation (tests/golden/g9_expected.npz).
        tests/test_model_gpu.py::test_resume_from_a_reference_written_checkpoint loads the file through this repo's
        `get_optimization(--resume)` on the GPU and must land on the same parameters / moments.
        Also written: g9_pretrained_keys.bin -- a state dict with EXACTLY the key set of the reference's
        `BertForMultiModalPreTraining` (what the public Conceptual-Captions `pretrained_model.bin` holds, vilbert.py:1119-1172),
        plus the missing / unexpected key lists the reference's own `Lily.from_pretrained` reports for it.
verify: gpurun_out/g9_repo_ckpt.bin (written by THIS repo's `save_model` after 2 HIP steps) is loaded into the reference through the
        reference's `get_optimization(args.resume)` (vilbert_init.py:44-70), one reference step is taken, and the result is compared
        with the repo's own third step (gpurun_out/g9_repo_step3.npz).  The verdict is stored in tests/golden/g9_interop_report.json.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))

import ref_import  # noqa: E402
from gen_golden import ZERO_DROP, build_lily, load_cfg, ref_args, ref_losses  # noqa: E402
from ytvln import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out")
BATCH = dict(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0)
SEED_W, LR, LOADER_LEN = 11, 1e-3, 10


class _Log:
    def info(self, *a, **k):
        pass


def make_args(**kw):
    return ref_args(ranking=True, traj_judge=True, masked_vision=True, masked_language=True, learning_rate=LR, **kw)


def ref_step(R, model, opt, sched, batch, args):
    model.train()
    outputs = model(*R.utils_init.get_model_input(batch))
    total, _ = ref_losses(R, batch, outputs, args)
    total.backward()
    opt.step()
    sched.step()
    model.zero_grad()
    return float(total)


def summarize(model, opt):
    out = {}
    for n, p in model.named_parameters():
        out["p/" + n] = p.detach().numpy().copy()
        st = opt.state.get(p, {})
        if "exp_avg" in st:
            out["m/" + n] = st["exp_avg"].numpy().copy()
            out["v/" + n] = st["exp_avg_sq"].numpy().copy()
            out["step/" + n] = np.array(st["step"])
    return out


def write(R):
    rcfg, _ = load_cfg(R, "micro.json", **ZERO_DROP)
    args = make_args()
    model, _, _ = build_lily(R, rcfg, args, seed=SEED_W)
    batch = synth.to_torch(synth.make_batch(**BATCH))
    opt, sched, _, _ = R.vilbert_init.get_optimization(args, model, LOADER_LEN, _Log())
    losses = [ref_step(R, model, opt, sched, batch, args) for _ in range(2)]
    with tempfile.TemporaryDirectory() as td:
        R.utils_init.save_model(td, "ckpt", _Log(), model, opt, sched, 4)          # the reference's writer, untouched
        blob = open(os.path.join(td, "ckpt.bin"), "rb").read()
    open(os.path.join(GOLD, "g9_ref_ckpt.bin"), "wb").write(blob)
    losses.append(ref_step(R, model, opt, sched, batch, args))
    exp = summarize(model, opt)
    exp["losses"] = np.array(losses)
    exp["lr_after"] = np.array(sched.get_last_lr())
    np.savez_compressed(os.path.join(GOLD, "g9_expected.npz"), **exp)
    print(f"g9_ref_ckpt.bin {len(blob)} bytes; losses {losses}")

    # the key set of the public pretrained_model.bin = BertForMultiModalPreTraining's state dict
    pcfg, _ = load_cfg(R, "micro.json", **ZERO_DROP)
    pre = R.vilbert.BertForMultiModalPreTraining(pcfg)
    shapes = {k: tuple(v.shape) for k, v in pre.state_dict().items()}
    W = synth.make_weights(shapes, 17)
    torch.save({k: torch.from_numpy(v) for k, v in W.items()}, os.path.join(GOLD, "g9_pretrained_keys.bin"))
    # what the reference itself does with such a file (Lily.from_pretrained, non-strict): which Lily tensors stay at their init values
    rcfg2, _ = load_cfg(R, "micro.json", **ZERO_DROP)
    rcfg2.args = make_args()
    torch.manual_seed(123)
    lily = R.lily.Lily.from_pretrained(os.path.join(GOLD, "g9_pretrained_keys.bin"), rcfg2, default_gpu=False)
    loaded = {k for k, v in lily.state_dict().items() if k in W and np.array_equal(v.numpy(), W[k])}
    report = {"pretrained_keys": sorted(shapes), "lily_keys_loaded_by_reference": sorted(loaded),
              "lily_keys_left_at_init_by_reference": sorted(set(lily.state_dict()) - loaded),
              "pretrained_keys_unused_by_reference": sorted(set(shapes) - set(lily.state_dict()))}
    json.dump(report, open(os.path.join(GOLD, "g9_pretrained_keys.json"), "w"), indent=0)
    print("pretrained key set:", len(shapes), "loaded", len(loaded), "left at init", len(report["lily_keys_left_at_init_by_reference"]),
          "unused", report["pretrained_keys_unused_by_reference"])


def verify(R):
    path = os.path.join(OUT, "g9_repo_ckpt.bin")
    want = np.load(os.path.join(OUT, "g9_repo_step3.npz"))
    rcfg, _ = load_cfg(R, "micro.json", **ZERO_DROP)
    args = make_args(resume=True, from_pretrained=path)
    model, _, _ = build_lily(R, rcfg, args, seed=SEED_W + 1)            # different initial weights: everything must come from the file
    opt, sched, _, start_epoch = R.vilbert_init.get_optimization(args, model, LOADER_LEN, _Log())      # the reference's resume path
    batch = synth.to_torch(synth.make_batch(**BATCH))
    loss = ref_step(R, model, opt, sched, batch, args)
    got = summarize(model, opt)
    worst = {"p": 0.0, "m": 0.0, "v": 0.0}
    for k, v in got.items():
        kind = k.split("/")[0]
        if kind in worst:
            worst[kind] = max(worst[kind], float(np.abs(v - want[k]).max()))
    steps_ok = all(int(got[k]) == 3 for k in got if k.startswith("step/"))
    ck = torch.load(path, map_location="cpu")
    report = {"direction": "checkpoint written by this repo's save_model on MI355X -> reference get_optimization(--resume) -> one reference step",
              "start_epoch": int(start_epoch), "reference_step3_loss": loss, "repo_step3_loss": float(want["loss3"]),
              "max_abs_diff_vs_repo_step3": worst, "optimizer_steps_all_3": bool(steps_ok),
              "checkpoint_keys": sorted(ck), "extra_keys_ignored_by_reference": sorted(set(ck) - {"model_state_dict", "optimizer_state_dict",
                                                                                                 "scheduler_state_dict", "epoch"}),
              "tolerance": {"p": 2e-6, "m": 1e-6, "v": 1e-8}}
    report["ok"] = bool(steps_ok and start_epoch == 5 and worst["p"] < 2e-6 and worst["m"] < 1e-6 and worst["v"] < 1e-8
                        and abs(loss - float(want["loss3"])) < 1e-4)
    json.dump(report, open(os.path.join(GOLD, "g9_interop_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))
    if not report["ok"]:
        raise SystemExit("repo-written checkpoint does NOT resume identically in the reference")


if __name__ == "__main__":
    R = ref_import.import_reference()
    torch.set_num_threads(8)
    mode = sys.argv[1] if len(sys.argv) > 1 else "write"
    {"write": write, "verify": verify}[mode](R)
