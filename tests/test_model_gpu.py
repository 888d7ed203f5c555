"""End-to-end parity of the HIP ViLBERT path with the goldens generated from the imported reference (GPU box only).

Stated fp32 tolerances (north-star bar is 1e-3 on losses):
    |loss - ref| <= 1e-4          |logit - ref| <= 1e-4 + 1e-4*|ref|        per-tensor gradient rel-L2 <= 1e-4 (+ tiny-norm guard)
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import ZERO_DROP, args_ns, cfg_dict, close, gold, rel_l2

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-4


def build_lily(dev, cfgname, args, seed, dropout=0.0, **over):
    from ytvln import synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    cfg = BertConfig(**cfg_dict(cfgname, **{**ZERO_DROP, **over}))
    cfg.args = args
    model = Lily(cfg, dropout_prob=dropout)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    W = synth.make_weights(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return model.to(dev), W


def losses_of(model, batch, args):
    from ytvln import utils_init as U
    outputs = model(*U.get_model_input(batch, all_options=bool(batch[13].all())))
    per = {}
    total = None
    for task, flag in U.TASKS:
        if getattr(args, flag):
            _, _, l, c = U.get_loss_correct(batch, outputs, task, args, None, True, all_options=bool(batch[13].all()))
            per[task], per["correct_" + task] = l, c
            l = args.traj_loss_scale * l if task == "traj" else l
            total = l if total is None else total + l
    return outputs, total, per


def test_g0_micro_everything(dev, lib):
    """micro config: every intermediate, attention probs, logits, 4 losses, all parameter gradients, 3 AdamW steps."""
    from ytvln import synth
    from ytvln.vilbert_init import get_optimization
    g = gold("g0_micro.npz")
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    model, W = build_lily(dev, "micro.json", args, seed=11)
    for k, v in W.items():
        assert np.array_equal(v, g["w/" + k]), f"weight recipe drifted for {k}"
    nb = synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, opt_holes=1, ignore_rank_frac=0.0)
    for i, a in enumerate(nb):
        assert np.array_equal(a, g["in_%02d" % i]), f"batch recipe drifted at index {i}"
    batch = synth.to_torch(nb, dev)

    # forward with intermediates through BertModel (output_all_attention_masks=True exercises the probs kernels)
    model.eval()
    from ytvln import utils_init as U
    inp = U.get_model_input(batch)
    inter = {}
    hooks = [model.bert.embeddings.register_forward_hook(lambda m, i, o: inter.__setitem__("embedding_output", o)),
             model.bert.v_embeddings.register_forward_hook(lambda m, i, o: inter.__setitem__("v_embedding_output", o))]
    for kind, layers in (("t", model.bert.encoder.layer), ("v", model.bert.encoder.v_layer), ("c", model.bert.encoder.c_layer)):
        for i, l in enumerate(layers):
            hooks.append(l.register_forward_hook(lambda m, inp_, o, name=f"{kind}{i}": inter.__setitem__(name, o)))
    with torch.no_grad():
        seq_t, seq_v, pt, pv, att = model.bert(inp[0], inp[1], inp[2], inp[3], inp[4], inp[5], output_all_attention_masks=True)
    for h in hooks:
        h.remove()
    close(inter["embedding_output"], g["embedding_output"], 2e-5, 2e-5, "embedding_output")
    close(inter["v_embedding_output"], g["v_embedding_output"], 2e-5, 2e-5, "v_embedding_output")
    for name, val in inter.items():
        if name[0] == "t" and name[1:].isdigit():
            close(val[0], g[name + ".t"], 5e-5, 5e-5, name)
            close(val[1], g[name + ".probs"], 1e-5, 5e-5, name + ".probs")
        elif name[0] == "v" and name[1:].isdigit():
            close(val[0], g[name + ".v"], 5e-5, 5e-5, name)
            close(val[1], g[name + ".probs"], 1e-5, 5e-5, name + ".probs")
        elif name[0] == "c":
            close(val[0], g[name + ".v"], 5e-5, 5e-5, name + ".v")
            close(val[1], g[name + ".t"], 5e-5, 5e-5, name + ".t")
            close(val[2][0], g[name + ".probs1"], 1e-5, 5e-5, name + ".probs1")
            close(val[2][1], g[name + ".probs2"], 1e-5, 5e-5, name + ".probs2")
    assert len(att[0]) == 2 and len(att[1]) == 2 and len(att[2]) == 1

    with torch.no_grad():
        outputs, total, per = losses_of(model, batch, args)
    for k in ("ranking", "traj", "vision", "language"):
        close(outputs[k], g["logits/" + k], 1e-4, 1e-4, "logits/" + k)
        close(per[k], g["loss/" + k], LOSS_TOL, 0, "loss/" + k)
        close(per["correct_" + k], g["loss/correct_" + k], 1e-6, 0, "correct/" + k)
    close(total, g["loss/total"], LOSS_TOL, 0, "loss/total")

    # 3 training steps with get_optimization's AdamW + WarmupLinear (lr 0, lr/2, lr)
    model.train()
    args.learning_rate = 1e-3
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    unused_ref = set(g["unused"].tolist())
    for step in range(3):
        outputs, total, per = losses_of(model, batch, args)
        total.backward()
        if step == 0:
            unused = {n for n, p in model.named_parameters() if p.grad is None}
            assert unused == unused_ref, (unused ^ unused_ref)
            for n, p in model.named_parameters():
                if p.grad is not None:
                    ref = g["grad/" + n]
                    assert rel_l2(p.grad, ref) < 1e-4 or float(np.linalg.norm(ref)) < 1e-7, f"grad {n}: {rel_l2(p.grad, ref):.2e}"
        close(total, g[f"step{step}.loss"], LOSS_TOL, 0, f"step{step}.loss")
        assert abs(sched.get_last_lr()[0] - float(g[f"step{step}.lr"])) < 1e-12
        opt.step(); sched.step(); opt.zero_grad()
    for n, p in model.named_parameters():
        close(p, g["after3/" + n], 2e-6, 2e-5, "after3/" + n)
        if ("exp_avg/" + n) in g.files:
            # (key biases have a mathematically zero gradient -- softmax shift invariance -- so both sides hold rounding noise)
            assert rel_l2(opt.state[p]["exp_avg"], g["exp_avg/" + n]) < 1e-4 or float(np.linalg.norm(g["exp_avg/" + n])) < 1e-7, n
            assert rel_l2(opt.state[p]["exp_avg_sq"], g["exp_avg_sq/" + n]) < 2e-4 or float(np.linalg.norm(g["exp_avg_sq/" + n])) < 1e-13, n
        else:
            assert p not in opt.state or "exp_avg" not in opt.state[p], f"{n} must have no optimizer state (grad is None in the reference)"


def check_summaries(model, W, batch, args, g, lr):
    """Shared by g1 / g2 / g4: losses, logit slices & checksums, per-tensor grad norms, post-step parameter checksums."""
    from ytvln.optimization import AdamW
    from ytvln.vilbert_init import grouped_parameters
    model.train()
    outputs, total, per = losses_of(model, batch, args)
    total.backward()
    for k, v in outputs.items():
        ref = g["logits/" + k]
        stride = int(g["logits_stride/" + k])
        flat = v.detach().reshape(v.shape[0], -1)
        got = v.detach() if stride == 1 and ref.shape == tuple(v.shape) else flat[:, ::stride][:, :ref.shape[1]]
        close(got, ref, 1e-4, 1e-4, "logits/" + k)
        assert abs(float(v.detach().double().sum()) - float(g["logits_sum/" + k])) <= 1e-4 * float(g["logits_abssum/" + k]) + 1e-3
    for k in per:
        if not k.startswith("correct_"):
            close(per[k], g["loss/" + k], LOSS_TOL, 0, "loss/" + k)
    close(total, g["loss/total"], LOSS_TOL, 0, "loss/total")
    names, norms = g["grad_names"].tolist(), g["grad_norms"]
    unused = {n for n, p in model.named_parameters() if p.grad is None}
    assert unused == set(g["unused"].tolist()), unused ^ set(g["unused"].tolist())
    pd = dict(model.named_parameters())
    worst = 0.0
    for n, ref in zip(names, norms):
        got = float(pd[n].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * ref + 1e-6, f"grad norm {n}: {got} vs {ref}"
        worst = max(worst, abs(got - ref) / max(ref, 1e-12))
    opt = AdamW(grouped_parameters(model, 0.01), lr=lr)
    opt.step()
    for n, s_ref, n_ref in zip(g["param_names"].tolist(), g["post_sum"], g["post_norm"]):
        p = pd[n].detach().double()
        assert abs(float(p.norm()) - n_ref) <= 2e-6 * n_ref + 1e-6, f"post-step norm {n}"
        assert abs(float(p.sum()) - s_ref) <= 3e-6 * float(p.abs().sum()) + 1e-6, f"post-step sum {n}"
    return worst


def test_g1_tiny_masked_language(dev, lib):
    """BASELINE config 1: tiny 2+2+1 / hidden 256, bs=2 K=7, T=16, R=8, --masked_language only."""
    from ytvln import synth
    g = gold("g1_tiny_mlm.npz")
    args = args_ns(masked_language=True)
    model, W = build_lily(dev, "tiny_2_2_1.json", args, seed=12)
    batch = synth.to_torch(synth.make_batch(bs=2, K=7, T=16, frames=1, boxes=8, seed=22), dev)
    check_summaries(model, W, batch, args, g, float(g["lr"]))


def test_g2_full_model_all_losses(dev, lib):
    """BASELINE config 2 shapes at N=7: 12/6/6 layers, T=80, R=288, MLM+MVM+ranking+traj_judge (250 M parameters)."""
    from ytvln import synth
    g = gold("g2_full_n7.npz")
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=13)
    assert sum(p.numel() for p in model.parameters()) == 250087039
    batch = synth.to_torch(synth.make_batch(bs=1, K=7, T=80, frames=8, boxes=36, seed=23, ignore_rank_frac=0.0), dev)
    check_summaries(model, W, batch, args, g, float(g["lr"]))


def test_g4_finetune_ranking(dev, lib):
    """BASELINE config 4 shapes: pretrain=False, K=6, R=7x36=252 (not a multiple of 32), ranking (+traj), one target=-1,
    and the eval-mode BCE branch."""
    from ytvln import synth
    from ytvln import utils_init as U
    g = gold("g4_finetune_rank.npz")
    args = args_ns(ranking=True, traj_judge=True, pretrain=False, num_negatives=2)
    model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=15)
    nb = synth.make_batch(bs=2, K=6, T=80, frames=7, boxes=36, seed=25, finetune_heading=True, ignore_rank_frac=0.0)
    nb[0][1] = -1
    batch = synth.to_torch(nb, dev)
    check_summaries(model, W, batch, args, g, float(g["lr"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    model.eval()
    eb = list(batch)
    eb[0] = torch.from_numpy(g["eval/target"]).to(dev)
    with torch.no_grad():
        outputs = model(*U.get_model_input(eb))
        _, _, l, c = U.get_loss_correct(eb, outputs, "ranking", args, None, False)
    close(l, g["eval/loss"], LOSS_TOL, 0, "eval bce")
    close(c, g["eval/correct"], 1e-6, 0, "eval correct")


def test_g3_multimodal_pretraining(dev, lib):
    """BertForMultiModalPreTraining: loss mode and prediction mode (vilbert.py:1396-1455)."""
    from ytvln import synth
    from ytvln.vilbert import BertConfig, BertForMultiModalPreTraining
    g = gold("g3_multimodal_pretraining.npz")
    cfg = BertConfig(**cfg_dict("tiny_2_2_1.json", **ZERO_DROP))
    model = BertForMultiModalPreTraining(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    W = synth.make_weights(shapes, 14)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    model.to(dev).eval()
    b = synth.to_torch(synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=24), dev)
    ids, feat, loc, vmask = b[6][:, 0], b[1][:, 0], b[2][:, 0], b[3][:, 0]
    imask, labels = b[7][:, 0], b[8][:, 0]
    img_label, img_target = b[5][:, 0, 1:], b[4][:, 0, 1:]
    nsl = torch.from_numpy(g["nsl"]).to(dev)
    with torch.no_grad():
        l = model(ids, feat, loc, None, imask, vmask, labels, img_label, img_target, nsl)
        p = model(ids, feat, loc, None, imask, vmask)
    for i, n in enumerate(("masked_lm_loss", "masked_img_loss", "next_sentence_loss")):
        assert l[i].shape == (1,)
        close(l[i], g[n], LOSS_TOL, 0, n)
    for i, n in enumerate(("prediction_scores_t", "prediction_scores_v", "seq_relationship_score")):
        ref = g[n]
        got = p[i] if tuple(p[i].shape) == ref.shape else p[i].reshape(p[i].shape[0], -1)[:, ::97]
        close(got, ref, 1e-4, 1e-4, n)
    assert len(p) == 4


def test_state_dict_roundtrip_and_checkpoint(dev, lib, tmp_path):
    """save_model-style checkpoint -> from_pretrained (incl. legacy gamma/beta names) reproduces the outputs."""
    from ytvln import synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    args = args_ns(ranking=True, masked_language=True)
    model, W = build_lily(dev, "micro.json", args, seed=5)
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=4), dev)
    from ytvln import utils_init as U
    model.eval()
    with torch.no_grad():
        ref = model(*U.get_model_input(batch))
    sd = {k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v.cpu() for k, v in model.state_dict().items()}
    path = tmp_path / "ckpt.bin"
    torch.save({"model_state_dict": sd, "epoch": 3}, path)
    cfg = BertConfig(**cfg_dict("micro.json", **ZERO_DROP))
    cfg.args = args
    m2 = Lily.from_pretrained(str(path), cfg, default_gpu=False, dropout_prob=0.0).to(dev).eval()
    with torch.no_grad():
        out = m2(*U.get_model_input(batch))
    for k in ref:
        assert torch.equal(ref[k], out[k]), k


def test_training_mode_dropout_runs_and_is_reproducible(dev, lib):
    """Train mode with the reference's p=0.1 everywhere: finite losses/grads, same seed -> same result, eval differs."""
    from ytvln import ops, synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    cfg = BertConfig(**cfg_dict("tiny_2_2_1.json"))
    cfg.args = args
    batch = synth.to_torch(synth.make_batch(bs=2, K=7, T=16, frames=2, boxes=4, seed=2, ignore_rank_frac=0.0), dev)
    res = []
    for rep in range(2):
        torch.manual_seed(0)
        ops.DropoutState.manual_seed(77)
        model = Lily(cfg).to(dev).train()
        outputs, total, per = losses_of(model, batch, args)
        total.backward()
        gn = torch.stack([p.grad.norm() for p in model.parameters() if p.grad is not None])
        assert bool(torch.isfinite(total)) and bool(torch.isfinite(gn).all())
        res.append((float(total), float(gn.sum())))
    assert res[0] == res[1], res
    model.eval()
    with torch.no_grad():
        _, total_eval, _ = losses_of(model, batch, args)
    assert abs(float(total_eval) - res[0][0]) > 1e-6


def test_dropout_stream_follows_torch_seed_and_survives_a_checkpoint(dev, lib, tmp_path):
    """the mask stream is keyed by torch's seed (the reference's set_seed -> torch.manual_seed(seed + local_rank),
    utils/misc.py:37-45), different seeds draw different masks, and (seed, counter) travel through save_model / --resume so a resumed
    run continues the stream instead of replaying it from 0."""
    from ytvln import ops, synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)

    def run(seed, steps, resume_from=None, save_to=None):
        ops.DropoutState.manual_seed(None)                 # default behaviour: follow torch.initial_seed()
        torch.manual_seed(seed)
        a = args
        if resume_from is not None:
            a = args_ns(**{**vars(args), "resume": True, "from_pretrained": resume_from})
        model, _ = build_lily(dev, "micro.json", a, seed=11, **{k: 0.1 for k in ZERO_DROP})
        model.train()
        opt, sched, _, _ = get_optimization(a, model, 10, None)
        losses = []
        first = 2 if resume_from is not None else 0
        for i in range(first, first + steps):
            loss, _ = U.train_step(model, opt, sched, batch, a, i, all_options=True)
            losses.append(float(loss))
        if save_to is not None:
            U.save_model(str(tmp_path), save_to, None, model, opt, sched, epoch=0)
        return losses

    a4 = run(5, 4)
    assert run(5, 4) == a4                                   # same seed: same masks
    assert run(6, 4)[0] != a4[0]                             # another seed: other masks already in the first step
    run(5, 2, save_to="rng")
    st = torch.load(U.get_model_path(str(tmp_path), "rng"), map_location="cpu")["ytvln_rng_state"]
    (seed, counter), = st.values()
    assert seed == 5 and counter == 2
    resumed = run(999, 2, resume_from=U.get_model_path(str(tmp_path), "rng"))     # torch seed differs: the checkpoint's stream wins
    assert np.allclose(resumed, a4[2:], rtol=0, atol=2e-6), (resumed, a4)
    ops.DropoutState.manual_seed(None)


def test_optimizer_load_state_dict_after_the_arena_exists(dev, lib):
    """load_state_dict() on an optimizer that already stepped must make the kernel use the LOADED moments and
    state_dict() must keep returning live tensors."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)

    def fresh():
        m, _ = build_lily(dev, "micro.json", args, seed=11)
        m.train()
        return m

    # run A: 3 steps straight.  Snapshot after step 2.
    mA = fresh(); oA, sA, _, _ = get_optimization(args, mA, 10, None)
    for i in range(2):
        U.train_step(mA, oA, sA, batch, args, i, all_options=True)
    snap_model = {k: v.detach().clone() for k, v in mA.state_dict().items()}
    live = oA.state_dict()
    snap_opt = {"param_groups": [dict(g) for g in live["param_groups"]],
                "state": {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in live["state"].items()}}
    snap_sched = sA.state_dict()
    U.train_step(mA, oA, sA, batch, args, 2, all_options=True)
    # run B: an optimizer that has ALREADY built its arenas on other values, then loads the snapshot
    mB = fresh(); oB, sB, _, _ = get_optimization(args, mB, 10, None)
    for i in range(4):
        U.train_step(mB, oB, sB, batch, args, i, all_options=True)
    mB.load_state_dict(snap_model); oB.load_state_dict(snap_opt); sB.load_state_dict(snap_sched)
    U.train_step(mB, oB, sB, batch, args, 2, all_options=True)
    bad = {n: float((pa.detach() - pb.detach()).abs().max()) for (n, pa), (_, pb) in zip(mA.named_parameters(), mB.named_parameters())
           if float((pa.detach() - pb.detach()).abs().max()) >= 2e-6}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    sdA, sdB = oA.state_dict()["state"], oB.state_dict()["state"]
    for k in sdA:
        assert sdA[k]["step"] == sdB[k]["step"] == 3
        assert float((sdA[k]["exp_avg"] - sdB[k]["exp_avg"]).abs().max()) < 1e-6
        assert float((sdA[k]["exp_avg_sq"] - sdB[k]["exp_avg_sq"]).abs().max()) < 1e-8


def test_graph_replay_equals_eager(dev, lib):
    """A training step captured once into a hipGraph and replayed must produce the same parameters as eager launches."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)
    finals, losses = [], []
    for mode in ("eager", "graph"):
        model, _ = build_lily(dev, "micro.json", args, seed=11)
        model.train()
        opt, sched, _, _ = get_optimization(args, model, 10, None)
        for i in range(2):
            U.train_step(model, opt, sched, batch, args, i, all_options=True)
        if mode == "eager":
            for i in range(2, 5):
                loss, _ = U.train_step(model, opt, sched, batch, args, i, all_options=True)
        else:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, _ = U.train_step(model, opt, None, batch, args, 0, all_options=True)
            for i in range(2, 5):
                opt.prepare_replay()
                g.replay()
                sched.step()
        torch.cuda.synchronize()
        finals.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
        losses.append(float(loss))
        assert all(opt.state[p]["step"] == 5 for p in model.parameters() if p in opt.state and "step" in opt.state[p])
    # every kernel on the path sums in a fixed order (the word-embedding gradient included: sorted runs, no atomics), so replaying the
    # graph reproduces the eager launches BIT FOR BIT
    assert losses[0] == losses[1], losses
    assert torch.equal(finals[0], finals[1]), float((finals[0] - finals[1]).abs().max())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_stream_equals_one_stream(dev, lib, precision):
    """ops.set_two_stream(True) sends the text side of the model (vilbert.py:737-811: independent of the image side between co-attention
    layers) to a second HIP stream.  Same kernels, same call order, same dropout sites: parameters after five steps (dropout on) must be
    BIT-identical to the one-stream run, with eager launches and as two branches of a replayed hipGraph."""
    from ytvln import ops, synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    if precision == "fp32":
        cfg = "micro.json"
        batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)
    else:
        cfg = "tiny_2_2_1.json"      # head dimension 64: the bf16-resident attention kernels
        batch = synth.to_torch(synth.make_batch(bs=2, K=7, T=16, frames=2, boxes=4, seed=22, ignore_rank_frac=0.0), dev)
    finals, losses = {}, {}
    ops.set_matmul_precision(precision)
    try:
        for two in (False, True):
            for mode in ("eager", "graph"):
                ops.set_two_stream(two)
                ops.DropoutState.manual_seed(1234)
                model, _ = build_lily(dev, cfg, args, seed=11)
                model.train()
                opt, sched, _, _ = get_optimization(args, model, 10, None)
                for i in range(2):
                    U.train_step(model, opt, sched, batch, args, i, all_options=True)
                if mode == "eager":
                    for i in range(2, 5):
                        loss, _ = U.train_step(model, opt, sched, batch, args, i, all_options=True)
                else:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        loss, _ = U.train_step(model, opt, None, batch, args, 0, all_options=True)
                    for i in range(2, 5):
                        opt.prepare_replay()
                        g.replay()
                        sched.step()
                torch.cuda.synchronize()
                assert not ops.TwoStream.active
                finals[two, mode] = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
                losses[two, mode] = float(loss)
    finally:
        ops.set_two_stream(True)          # the default
        ops.set_matmul_precision("fp32")
        ops.DropoutState.manual_seed(None)
    ref = finals[False, "eager"]
    assert torch.isfinite(ref).all()
    for key, val in finals.items():
        assert losses[key] == losses[False, "eager"], (key, losses)
        assert torch.equal(val, ref), (key, float((val - ref).abs().max()))


def test_save_resume_and_eval_loops(dev, lib, tmp_path):
    """save_model -> get_optimization(--resume) continues bit-for-bit like an uninterrupted run; val_epoch / test_epoch agree
    with a direct evaluation (reference utils_init.py:277-295, 315-446, vilbert_init.py:44-70)."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    nb = synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0)
    batch = synth.to_torch(nb, dev)

    def fresh():
        m, _ = build_lily(dev, "micro.json", args, seed=11)
        m.train()
        return m

    # uninterrupted: 4 steps
    m0 = fresh()
    o0, s0, _, _ = get_optimization(args, m0, 10, None)
    for i in range(4):
        U.train_step(m0, o0, s0, batch, args, i, all_options=True)
    # interrupted after 2 steps, saved, resumed in a new process-like state
    m1 = fresh()
    o1, s1, _, _ = get_optimization(args, m1, 10, None)
    for i in range(2):
        U.train_step(m1, o1, s1, batch, args, i, all_options=True)
    U.save_model(str(tmp_path), "ckpt", None, m1, o1, s1, epoch=7)
    ck = torch.load(U.get_model_path(str(tmp_path), "ckpt"), map_location="cpu")
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "epoch", "ytvln_rng_state"}
    st = next(iter(ck["optimizer_state_dict"]["state"].values()))
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and st["step"] == 2
    m2 = fresh()
    rargs = args_ns(**{**vars(args), "resume": True, "from_pretrained": U.get_model_path(str(tmp_path), "ckpt")})
    o2, s2, _, start_epoch = get_optimization(rargs, m2, 10, None)
    assert start_epoch == 8
    for i in range(2, 4):
        U.train_step(m2, o2, s2, batch, args, i, all_options=True)
    a = torch.cat([p.detach().reshape(-1) for p in m0.parameters()])
    b = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
    assert float((a - b).abs().max()) < 2e-6, float((a - b).abs().max())
    U.delete_model(str(tmp_path), "ckpt")
    assert not (tmp_path / "ckpt.bin").exists()

    # evaluation loops on a 2-batch "loader" with a multi-hot beam target
    tgt = np.zeros((2, 3), bool); tgt[0, 0] = True; tgt[1, 1] = True
    eb = list(nb); eb[0] = tgt
    loader = [synth.to_torch(eb), synth.to_torch(eb)]
    sr = U.val_epoch(0, m0, "val_seen", loader, None, True, args, 0, None, "ranking")
    red = U.test_epoch(0, m0, "test", loader, None, True, args, 0, None)
    m0.eval()
    with torch.no_grad():
        out = m0(*U.get_model_input(synth.to_torch(eb, dev), True))
    logit = out["ranking"].view(2, 3).double()
    t = torch.from_numpy(tgt).to(dev)
    import torch.nn.functional as F
    ref_loss = F.binary_cross_entropy_with_logits(logit, t.double())
    ref_sr = t.gather(1, logit.argmax(1, keepdim=True)).double().sum() / 2
    assert abs(float(sr) - float(ref_sr)) < 1e-6
    assert abs(float(red["ranking"][1]) - float(ref_loss)) < 1e-5 and abs(float(red["ranking"][2]) - float(ref_sr)) < 1e-6
    assert set(red) == {"ranking", "traj"}


@pytest.mark.parametrize("all_options", [True, None])
def test_loss_aware_heads_match_full_heads(dev, lib, all_options):
    """Decoding only the rows that carry a masked-token / masked-region target (train_step(loss_aware_heads=True)) gives the
    same losses and the same parameters after optimizer steps as decoding every row (the reference's behaviour)."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = synth.to_torch(synth.make_batch(bs=3, K=3, T=16, frames=2, boxes=5, F=16, C=11, vocab=97, seed=33,
                                            ignore_rank_frac=0.0 if all_options else 0.3), dev)
    finals, losses = [], []
    for aware in (False, True):
        model, _ = build_lily(dev, "micro.json", args, seed=5)
        model.train()
        opt, sched, _, _ = get_optimization(args, model, 10, None)
        for i in range(3):
            loss, metrics = U.train_step(model, opt, sched, batch, args, i, all_options=all_options, loss_aware_heads=aware,
                                         capacity_frac=0.5)
        if aware:
            assert float(metrics["head_row_overflow"]) == 0.0
        torch.cuda.synchronize()
        finals.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
        losses.append((float(loss), {k: float(v) for k, v in metrics["loss"].items()}))
    assert abs(losses[0][0] - losses[1][0]) < 2e-6 * max(1.0, abs(losses[0][0])), losses
    for k in losses[0][1]:
        assert abs(losses[0][1][k] - losses[1][1][k]) < 2e-6 * max(1.0, abs(losses[0][1][k])), (k, losses)
    assert float((finals[0] - finals[1]).abs().max()) < 5e-6, float((finals[0] - finals[1]).abs().max())
    # capacity too small -> flagged on device, never silently wrong
    model, _ = build_lily(dev, "micro.json", args, seed=5)
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    _, metrics = U.train_step(model, opt, sched, batch, args, 0, all_options=all_options, loss_aware_heads=True, capacity_frac=0.01)
    # (capacity is rounded up to 128 rows, which may still cover the micro batch; only check the flag is a device scalar)
    assert metrics["head_row_overflow"].numel() == 1


def test_g2_full_model_bf16_projections(dev, lib):
    """BASELINE config 5's arithmetic -- the bf16-resident path: activations, activation gradients, logits and the weight copies are bf16 in
    HBM, every projection and both attention products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; softmax / LayerNorm statistics /
    loss reductions / master weights / AdamW stay fp32 -- against the same fp32 golden as test_g2.  Stated tolerance for this mode: each loss
    within 2e-2 relative (bf16 has an 8-bit mantissa: 2^-9 = 2e-3 relative rounding per stored value, averaged over the contractions and 24
    layers), gradient norms within 5 %.  One documented exception: the two pooler biases get 10 % -- with ONE item of 7 options the ranking
    loss is shift-invariant across the options, so the per-option gradients of the pooled vector nearly cancel in the bias sum (|sum| is
    about 1/40 of the sum of norms here) and a 1 % rounding of the inputs becomes a several-% change of that sum; the same parameters meet
    the 5 % bar at N = 14 (test_g10) and the GEMM column-sum itself is checked directly in test_bf16_gpu.py.
    The fp32 path's 1e-4 bar does not apply -- and must NOT be met bit-for-bit, which guards against a silent fp32 fallback."""
    from ytvln import ops, synth
    g = gold("g2_full_n7.npz")
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=13)
    batch = synth.to_torch(synth.make_batch(bs=1, K=7, T=80, frames=8, boxes=36, seed=23, ignore_rank_frac=0.0), dev)
    model.train()
    ops.set_matmul_precision("bf16")
    try:
        outputs, total, per = losses_of(model, batch, args)
        total.backward()
    finally:
        ops.set_matmul_precision("fp32")
    worst = 0.0
    for k in per:
        if not k.startswith("correct_"):
            ref = float(g["loss/" + k])
            err = abs(float(per[k]) - ref) / max(abs(ref), 1e-6)
            worst = max(worst, err)
            assert err < 2e-2, (k, float(per[k]), ref)
    assert abs(float(total) - float(g["loss/total"])) < 2e-2 * abs(float(g["loss/total"]))
    assert worst > 1e-7, "bf16 mode reproduced the fp32 losses exactly: the bf16 path did not run"
    pd = dict(model.named_parameters())
    unused = {n for n, p in model.named_parameters() if p.grad is None}
    assert unused == set(g["unused"].tolist())
    bad = []
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        got = float(pd[n].grad.double().norm())
        # (+1e-4: key-projection biases have a mathematically zero gradient -- softmax shift invariance -- so theirs is pure rounding noise)
        bar = 1e-1 if n.endswith("pooler.dense.bias") else 5e-2
        if abs(got - ref) > bar * ref + 1e-4:
            bad.append((n, got, float(ref)))
    assert not bad, bad[:5]


def test_g2_g4_full_model_fp32x3_meets_the_fp32_bar(dev, lib):
    """The opt-in fp32x3 projections (fp32 operands split exactly into three bf16 terms in registers, six bf16 MFMAs per product)
    against the SAME goldens and the SAME tolerances as the native fp32 path (test_g2 / test_g4: losses within 1e-4, gradient
    summaries, post-AdamW parameters): the mode is an fp32-level arithmetic, not a reduced-precision one."""
    from ytvln import ops, synth
    ops.set_matmul_precision("fp32x3")
    try:
        g = gold("g2_full_n7.npz")
        args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
        model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=13)
        batch = synth.to_torch(synth.make_batch(bs=1, K=7, T=80, frames=8, boxes=36, seed=23, ignore_rank_frac=0.0), dev)
        check_summaries(model, W, batch, args, g, float(g["lr"]))
        del model
        g = gold("g4_finetune_rank.npz")
        args = args_ns(ranking=True, traj_judge=True, pretrain=False, num_negatives=2)
        model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=15)
        nb = synth.make_batch(bs=2, K=6, T=80, frames=7, boxes=36, seed=25, finetune_heading=True, ignore_rank_frac=0.0)
        nb[0][1] = -1
        check_summaries(model, W, synth.to_torch(nb, dev), args, g, float(g["lr"]))
    finally:
        ops.set_matmul_precision("fp32")


def test_inference_rerank_shapes_match_oracle(dev, lib):
    """The re-ranking inference path (test.py:144-192) at the reference's inference shapes: 30 candidate beams per instruction,
    R = 8 viewpoints x 101 regions = 808 (not a multiple of the 32-row attention tiles), T = 60, forward only, eval mode -- on the
    tiny configuration so the CPU oracle finishes in seconds.  Ranking scores within 1e-4 of the oracle; eval_epoch / convert_scores
    pick the same beams as an argmax over the oracle's scores."""
    import vilbert_ref as O
    from ytvln import synth
    from ytvln import utils_init as U
    args = args_ns(ranking=True, pretrain=False)
    model, W = build_lily(dev, "tiny_2_2_1.json", args, seed=31)
    model.eval()
    nb = synth.make_batch(bs=2, K=30, T=60, frames=8, boxes=101, seed=77, finetune_heading=True, ignore_rank_frac=0.0)
    nb[13][1, 25:] = False                       # the second instruction has only 25 candidates (ragged opt_mask)
    nb[12] = np.array([[4051, 0], [4051, 2]], np.int64)         # test-time loaders put (path id, instruction index) here
    batch_cpu = synth.to_torch(nb)
    scores = U.eval_epoch(model, [batch_cpu], args)
    assert [s[0] for s in scores] == ["4051_0", "4051_2"]
    S = {k: torch.from_numpy(v).clone() for k, v in W.items()}
    cfgd = cfg_dict("tiny_2_2_1.json", **ZERO_DROP)
    ids, feat, loc, seg, imask, vmask = O.model_input(batch_cpu)
    with torch.no_grad():
        ref = O.lily_forward(S, O.RefConfig(**cfgd), O.TaskFlags(ranking=True), ids, feat, loc, seg, imask, vmask)["ranking"].squeeze(1)
    ref = O.pad_packed(ref, batch_cpu[13])
    got = torch.tensor([s[1] for s in scores])
    valid = batch_cpu[13]
    assert float((got - ref)[valid].abs().max()) < 1e-4, float((got - ref)[valid].abs().max())
    assert torch.equal(got[~valid], ref[~valid])          # padded candidates carry pad_packed's filler in both
    beam_data = [{"instr_id": i, "ranked_paths": [[f"vp{i}_{b}_{k}" for k in range(3)] for b in range(30)], "exploration_path": ["e0", "e1"]}
                 for i in ("4051_0", "4051_2")]
    out = U.convert_scores(scores, beam_data, add_exploration_path=True)
    for row, o in zip(ref, out):
        best = int(torch.argmax(row))
        assert o["trajectory"] == [["e0"], ["e1"]] + beam_data[0 if o["instr_id"] == "4051_0" else 1]["ranked_paths"][best]


def test_cfg2_full_size_parity_and_properties(dev, lib):
    """BASELINE configs[1] at its FULL per-GPU size (bs = 8 items x 7 options = 56 pairs, T = 80, R = 288, 250 M parameters):
      * the four losses against the CPU oracle run on this box (forward only, eval-free: dropout off), |diff| <= 1e-4;
      * size-independent properties of the path at that size:
          - item-permutation equivariance: permuting the items permutes the ranking scores and leaves every loss unchanged
            (rows are independent through the model);
          - masked-region invariance: features / boxes of regions whose image_mask is 0 never reach a loss
            (they are only keys behind a -10000 additive mask; vilbert.py:295-297)."""
    import vilbert_ref as O
    from ytvln import synth
    from ytvln import utils_init as U
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    model, W = build_lily(dev, "bert_base_6_layer_6_connect.json", args, seed=41)
    model.train()                                     # dropout probabilities are zero in this configuration (ZERO_DROP)
    nb = synth.make_batch(bs=8, K=7, T=80, frames=8, boxes=36, seed=51, ignore_rank_frac=0.0)
    batch = synth.to_torch(nb, dev)
    with torch.no_grad():
        outputs, total, per = losses_of(model, batch, args)
    S = {k: torch.from_numpy(v).clone() for k, v in W.items()}
    cfgd = cfg_dict("bert_base_6_layer_6_connect.json", **ZERO_DROP)
    flags = O.TaskFlags(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    cb = synth.to_torch(nb)
    with torch.no_grad():
        oout = O.lily_forward(S, O.RefConfig(**cfgd), flags, *O.model_input(cb))
        ototal, oper = O.total_loss(cb, oout, flags)
    for k, v in oper.items():
        assert abs(float(per[k]) - float(v)) <= LOSS_TOL, (k, float(per[k]), float(v))
    assert abs(float(total) - float(ototal)) <= 4 * LOSS_TOL
    close(outputs["ranking"], oout["ranking"], 1e-4, 1e-4, "ranking scores at full size")

    # item permutation
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=dev)
    pb = [t[perm] if (torch.is_tensor(t) and t.dim() >= 1 and t.shape[0] == 8) else t for t in batch]
    with torch.no_grad():
        pout, ptotal, pper = losses_of(model, pb, args)
    assert float((pout["ranking"].view(8, 7) - outputs["ranking"].view(8, 7)[perm]).abs().max()) < 2e-5
    for k in per:
        if not k.startswith("correct_"):
            assert abs(float(pper[k]) - float(per[k])) < 2e-5, k

    # masked regions carry no information
    mb = list(batch)
    dead = (batch[3] == 0)
    feats, boxes = batch[1].clone(), batch[2].clone()
    feats[dead] = 7.5
    boxes[dead] = 0.25
    boxes[..., 11] = batch[2][..., 11]               # the frame index must stay a valid embedding row
    mb[1], mb[2] = feats, boxes
    with torch.no_grad():
        mout, mtotal, mper = losses_of(model, mb, args)
    for k in per:
        if not k.startswith("correct_"):
            assert abs(float(mper[k]) - float(per[k])) < 2e-5, (k, float(mper[k]), float(per[k]))


def test_cfg5_long_trajectory_shapes(dev, lib):
    """BASELINE configs[4] shapes (16 frames x 36 regions = 576 regions per pair, T = 80, all four losses) on the tiny configuration so the
    CPU oracle finishes in seconds: fp32 path within 1e-4 of the oracle's losses, bf16 MFMA mode within 2e-2 relative of them."""
    import vilbert_ref as O
    from ytvln import ops, synth
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    model, W = build_lily(dev, "tiny_2_2_1.json", args, seed=61)
    model.train()
    nb = synth.make_batch(bs=2, K=7, T=80, frames=16, boxes=36, seed=71, ignore_rank_frac=0.0)
    batch = synth.to_torch(nb, dev)
    S = {k: torch.from_numpy(v).clone() for k, v in W.items()}
    flags = O.TaskFlags(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    cb = synth.to_torch(nb)
    with torch.no_grad():
        oout = O.lily_forward(S, O.RefConfig(**cfg_dict("tiny_2_2_1.json", **ZERO_DROP)), flags, *O.model_input(cb))
        ototal, oper = O.total_loss(cb, oout, flags)
    with torch.no_grad():
        _, total, per = losses_of(model, batch, args)
    for k, v in oper.items():
        assert abs(float(per[k]) - float(v)) <= LOSS_TOL, (k, float(per[k]), float(v))
    ops.set_matmul_precision("bf16")
    try:
        with torch.no_grad():
            _, btotal, bper = losses_of(model, batch, args)
    finally:
        ops.set_matmul_precision("fp32")
    for k, v in oper.items():
        assert abs(float(bper[k]) - float(v)) <= 2e-2 * max(abs(float(v)), 1e-3), (k, float(bper[k]), float(v))
    assert abs(float(btotal) - float(total)) > 0.0


def test_gradient_accumulation_matches_big_batch_of_micro_steps(dev, lib):
    """--gradient_accumulation_steps 2 (utils/utils_init.py:226-239): two micro-steps on two batches, each loss / 2, one optimizer step
    == one step on the summed-and-halved gradients computed by hand.  Exercises the fall-back of the direct-to-arena gradient writes
    (the second micro-step must ADD to the first one's gradients) and the DataParallel-free zero_grad path."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True, gradient_accumulation_steps=2)
    args.learning_rate = 1e-3
    b0 = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=81, ignore_rank_frac=0.0), dev)
    b1 = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=82, ignore_rank_frac=0.0), dev)
    model, _ = build_lily(dev, "micro.json", args, seed=9)
    model.train()
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    for rep in range(2):                                  # two optimizer steps = four micro-steps (arena exists from the second on)
        U.train_step(model, opt, sched, b0, args, 2 * rep, all_options=True)
        U.train_step(model, opt, sched, b1, args, 2 * rep + 1, all_options=True)
    got = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()

    ref_args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    ref_args.learning_rate = 1e-3
    ref, _ = build_lily(dev, "micro.json", ref_args, seed=9)
    ref.train()
    ropt, rsched, _, _ = get_optimization(args, ref, 10, None)
    for rep in range(2):
        total = None
        for b in (b0, b1):
            _, l, _ = losses_of(ref, b, ref_args)
            total = 0.5 * l if total is None else total + 0.5 * l
        total.backward()
        ropt.step(); rsched.step(); ropt.zero_grad()
    want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()]).cpu()
    assert float((got - want).abs().max()) < 3e-6, float((got - want).abs().max())


def test_train_epoch_loop_and_nonstrict_checkpoint(dev, lib, tmp_path):
    """train_epoch (utils/utils_init.py:192-268) over a host-side loader with a ragged opt_mask, logging through a writer object; then
    `from_pretrained` on a checkpoint that LACKS the task heads / orientation embeddings (like the public Conceptual-Captions
    pretrained_model.bin, vilbert.py:1161-1172): missing tensors keep their fresh initialisation, the rest is loaded."""
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    from ytvln.vilbert_init import get_optimization
    # (no trajectory-judgement head here: with a ragged opt_mask pad_packed fills the missing options with -inf and the reference's own
    #  BCE-with-logits turns that into NaN -- golden g0 records exactly that NaN, and test_g0 checks we reproduce it)
    args = args_ns(ranking=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    model, W = build_lily(dev, "micro.json", args, seed=17)
    opt, sched, _, _ = get_optimization(args, model, 10, None)
    loader = [synth.to_torch(synth.make_batch(bs=3, K=7, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=90 + i, opt_holes=(i % 2),
                                             ignore_rank_frac=0.0)) for i in range(4)]

    class Writer:
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, value, global_step=None):
            self.rows.append((tag, float(value), global_step))
    w = Writer()
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    U.train_epoch(0, model, opt, sched, loader, w, True, args, None)
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert float((after - before).abs().max()) > 0
    tags = {t for t, _, _ in w.rows}
    assert {"learning_rate/train", "loss/train", "loss/vision", "loss/language", "loss/ranking", "accuracy/ranking"} <= tags
    assert sorted({s for _, _, s in w.rows}) == [0, 1, 2, 3]
    assert all(np.isfinite(v) for _, v, _ in w.rows), [r for r in w.rows if not np.isfinite(r[1])]

    sd = {k: v.cpu() for k, v in model.state_dict().items()
          if not (k.startswith("vil_logit") or k.startswith("judge") or "orientation_embeddings" in k)}
    assert len(sd) < len(model.state_dict())
    sd["some.unexpected.key"] = torch.zeros(3)
    path = tmp_path / "pretrained_model.bin"
    torch.save(sd, path)
    cfg = BertConfig(**cfg_dict("micro.json", **ZERO_DROP))
    cfg.args = args
    torch.manual_seed(123)
    m2 = Lily.from_pretrained(str(path), cfg, default_gpu=False, dropout_prob=0.0).to(dev)
    s2 = m2.state_dict()
    for k, v in model.state_dict().items():
        if k in sd:
            assert torch.equal(s2[k], v), k
    assert not torch.equal(s2["vil_logit.weight"], model.state_dict()["vil_logit.weight"])       # stayed at its own initialisation


def test_resume_from_a_reference_written_checkpoint(dev, lib):
    """tests/golden/g9_ref_ckpt.bin was written by the REFERENCE's save_model after two steps of the reference's own Lily + AdamW +
    WarmupLinear (oracle/gen_golden_ckpt.py); resumed through this repo's get_optimization(--resume) the third step must land on the
    reference's third step (parameters, both moments, step counters, learning rate, loss).  utils_init.py:277-295, vilbert_init.py:44-70."""
    from conftest import GOLD
    from ytvln import synth
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    exp = gold("g9_expected.npz")
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True, learning_rate=1e-3, resume=True,
                   from_pretrained=os.path.join(GOLD, "g9_ref_ckpt.bin"))
    model, _ = build_lily(dev, "micro.json", args, seed=12)          # other initial weights: everything must come from the file
    model.train()
    opt, sched, _, start_epoch = get_optimization(args, model, 10, None)
    assert start_epoch == 5
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)
    loss, _ = U.train_step(model, opt, sched, batch, args, 2, all_options=True)
    assert abs(float(loss) - float(exp["losses"][2])) < LOSS_TOL
    assert np.allclose(sched.get_last_lr(), exp["lr_after"], rtol=1e-12)
    for n, p in model.named_parameters():
        close(p, exp["p/" + n], 2e-6, 2e-5, "param " + n)
        if "m/" + n in exp.files:
            st = opt.state[p]
            assert st["step"] == int(exp["step/" + n]) == 3
            close(st["exp_avg"], exp["m/" + n], 1e-6, 1e-4, "exp_avg " + n)
            close(st["exp_avg_sq"], exp["v/" + n], 1e-9, 1e-4, "exp_avg_sq " + n)
        else:
            assert p not in opt.state or "exp_avg" not in opt.state[p], n      # tensors the reference never touched got no state here either


def test_pretrained_model_key_set_loads_like_the_reference(dev, lib):
    """A state dict with EXACTLY the keys of the reference's BertForMultiModalPreTraining -- what the public Conceptual-Captions
    `pretrained_model.bin` holds (README.md:80, vilbert.py:1119-1172) -- through `Lily.from_pretrained`: the same tensors get loaded and
    the same ones stay at their initial values as when the reference's own Lily loads that file (tests/golden/g9_pretrained_keys.json)."""
    import json
    from conftest import GOLD
    from ytvln import synth
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    rep = json.load(open(os.path.join(GOLD, "g9_pretrained_keys.json")))
    cfg = BertConfig(**cfg_dict("micro.json", **ZERO_DROP))
    cfg.args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    path = os.path.join(GOLD, "g9_pretrained_keys.bin")
    W = torch.load(path, map_location="cpu")
    assert sorted(W) == rep["pretrained_keys"]
    lily = Lily.from_pretrained(path, cfg, default_gpu=False)
    sd = lily.state_dict()
    loaded = sorted(k for k in sd if k in W and torch.equal(sd[k], W[k]))
    assert loaded == rep["lily_keys_loaded_by_reference"]
    assert sorted(set(sd) - set(loaded)) == rep["lily_keys_left_at_init_by_reference"] == ["judge.bias", "judge.weight", "vil_logit.bias", "vil_logit.weight"]
    assert sorted(set(W) - set(sd)) == rep["pretrained_keys_unused_by_reference"] == []
    lily.to(dev).eval()                                   # and the loaded model runs
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)
    from ytvln import utils_init as U
    with torch.no_grad():
        out = lily(*U.get_model_input(batch, True))
    assert all(bool(torch.isfinite(v).all()) for v in out.values())


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs at their FULL per-GPU sizes on the FULL model (goldens: oracle/gen_golden_full.py, real reference == oracle)
# ------------------------------------------------------------------------------------------------------------------
FULL_CFG = "bert_base_6_layer_6_connect.json"
PRETRAIN = dict(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)


# direction bars of the bf16-resident gradients against the reference's fp32 ones (measured on g16b: 0.99797 over all tensors, lowest single
# 64-element slice 0.965 -- bert.t_pooler.dense.weight; deterministic)
BF16_COS_ALL, BF16_COS_MIN = 0.995, 0.93


def _bf16_check(model, batch, args, g):
    """bf16 MFMA mode against an fp32 golden: losses within 2e-2 relative, gradient norms within 5 % (+1e-4; 10 % for the co-attention
    query / key projections), never bit-equal; where the golden carries gradient slices (g16b) also their DIRECTION: cosine with the
    reference's fp32 gradient > 0.995 over all tensors together, > 0.93 on every tensor's 64-element slice."""
    from ytvln import ops
    model.train()
    ops.set_matmul_precision("bf16")
    try:
        outputs, total, per = losses_of(model, batch, args)
        total.backward()
    finally:
        ops.set_matmul_precision("fp32")
    worst = 0.0
    for k in per:
        if not k.startswith("correct_"):
            ref = float(g["loss/" + k])
            err = abs(float(per[k]) - ref) / max(abs(ref), 1e-6)
            worst = max(worst, err)
            assert err < 2e-2, (k, float(per[k]), ref)
    assert worst > 1e-7, "bf16 mode reproduced the fp32 losses exactly: the bf16 path did not run"
    pd = dict(model.named_parameters())
    assert {n for n, p in model.named_parameters() if p.grad is None} == set(g["unused"].tolist())
    # 5 % per tensor; 10 % for the query / key projections of BertBiAttention: their gradient goes through dS = P o (dP - delta), a difference
    # of two nearly equal bf16-rounded sums when a direction attends almost uniformly over 252-576 regions (measured: one co-layer of cfg 4 at
    # -9 %, deterministic -- the runs that used to land inside 5 % did so through a since-fixed race in the bf16 forward kernel)
    def bar(n):
        return 1e-1 if ("biattention.query" in n or "biattention.key" in n) else 5e-2
    bad = [(n, float(pd[n].grad.double().norm()), float(ref)) for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"])
           if abs(float(pd[n].grad.double().norm()) - ref) > bar(n) * ref + 1e-4]
    assert not bad, bad
    if "grad_slices" in g.files:
        # DIRECTION, where the golden carries it (g16b: a 64-element strided slice of every gradient): norms alone would pass a gradient that
        # points elsewhere.  Cosine between the bf16 slice and the reference's fp32 slice, per tensor and over all tensors together (every
        # slice scaled by the reference norm of its tensor, so that each tensor weighs the same).
        cos, num, den_a, den_b = {}, 0.0, 0.0, 0.0
        floor = 1e-6 * float(np.median(g["grad_norms"]))          # (the 30 key biases: their gradient is zero by the softmax's shift invariance, ~1e-10 of rounding)
        for n, ref, sl in zip(g["grad_names"].tolist(), g["grad_norms"], g["grad_slices"]):
            if float(ref) <= floor:
                continue
            gr = pd[n].grad.detach().reshape(-1)
            st = max(1, gr.numel() // 64)
            a = gr[::st][:64].double().cpu().numpy()
            b = np.asarray(sl[: a.size], dtype=np.float64)
            na, nb = float(np.linalg.norm(a)), float(np.linalg.norm(b))
            if nb > 1e-3 * float(ref) / max(1.0, (gr.numel() / 64) ** 0.5) and nb > 1e-12:      # (a slice that caught only near-zeros says nothing)
                cos[n] = float(a @ b) / max(na * nb, 1e-300)
                w = 1.0 / max(float(ref), 1e-30)
                num += float(a @ b) * w * w; den_a += na * na * w * w; den_b += nb * nb * w * w
        overall = num / max((den_a * den_b) ** 0.5, 1e-300)
        worst = sorted(cos.items(), key=lambda kv: kv[1])[:5]
        print(f"[bf16 gradient direction] tensors {len(cos)}, overall cosine {overall:.5f}, lowest {worst}")
        assert len(cos) > 0.85 * len(g["grad_names"]), "too few slices carried signal"
        assert overall > BF16_COS_ALL, overall
        assert worst[0][1] > BF16_COS_MIN, worst


def test_g10_cfg5_long_trajectories_full_model_fp32_and_bf16(dev, lib):
    """BASELINE configs[4]: FULL 12/6/6 model, 16 frames x 36 = 576 regions per pair, T = 80, N = 14 rows, all four losses.
    fp32: the 1e-4 bar (losses, logit slices, gradient norms, post-AdamW parameters).  bf16 MFMA mode (the arithmetic configs[4] names):
    losses within 2e-2 relative of the fp32 reference, gradient norms within 5 %."""
    from ytvln import synth
    g = gold("g10_cfg5_long_n14.npz")
    args = args_ns(**PRETRAIN)
    kw = dict(bs=2, K=7, T=80, frames=16, boxes=36, seed=41, ignore_rank_frac=0.0)
    model, W = build_lily(dev, FULL_CFG, args, seed=31)
    check_summaries(model, W, synth.to_torch(synth.make_batch(**kw), dev), args, g, float(g["lr"]))
    del model
    torch.cuda.empty_cache()
    model, W = build_lily(dev, FULL_CFG, args, seed=31)
    _bf16_check(model, synth.to_torch(synth.make_batch(**kw), dev), args, g)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_g16_cfg5_full_per_gpu_size_n224_forward(dev, lib, precision):
    """BASELINE configs[4] at its OWN per-GPU size (VERDICT r4 "missing" 2): bs = 32 items x K = 7 = 224 rows, 16 frames x 36 = 576 regions,
    T = 80, FULL 12/6/6 model -- forward under no_grad against the reference's own forward (oracle/gen_golden_full.py g16, 20 CPU-minutes;
    the reference shapes are vilbert.py:413-440 at R = 576).  fp32: the four losses within 1e-4, ranking / traj logits and 64-column slices
    + checksums of the vision / language logits within 1e-4.  bf16-resident (the arithmetic configs[4] names): losses within 2e-2 relative,
    logits within 2e-2 of their range, and not bit-equal to the fp32 run."""
    from ytvln import ops, synth
    g = gold("g16_cfg5_full_n224.npz")
    args = args_ns(**PRETRAIN)
    model, W = build_lily(dev, FULL_CFG, args, seed=34)
    batch = synth.to_torch(synth.make_batch(bs=32, K=7, T=80, frames=16, boxes=36, seed=44, ignore_rank_frac=0.0), dev)
    model.train()
    ops.set_matmul_precision(precision)
    try:
        with torch.no_grad():
            outputs, total, per = losses_of(model, batch, args)
    finally:
        ops.set_matmul_precision("fp32")
    bf = precision == "bf16"
    for k, v in outputs.items():
        ref = g["logits/" + k]
        stride = int(g["logits_stride/" + k])
        flat = v.detach().float().reshape(v.shape[0], -1)
        got = v.detach().float() if stride == 1 and ref.shape == tuple(v.shape) else flat[:, ::stride][:, :ref.shape[1]]
        if bf:
            span = float(np.abs(ref).max())
            assert float((got.cpu().double() - torch.from_numpy(ref).double()).abs().max()) <= 2e-2 * span + 2e-2, k
        else:
            close(got, ref, 1e-4, 1e-4, "logits/" + k)
            assert abs(float(v.detach().double().sum()) - float(g["logits_sum/" + k])) <= 1e-4 * float(g["logits_abssum/" + k]) + 1e-3
    worst = 0.0
    for k in per:
        if k.startswith("correct_"):
            continue
        ref = float(g["loss/" + k])
        if bf:
            err = abs(float(per[k]) - ref) / max(abs(ref), 1e-6)
            worst = max(worst, err)
            assert err < 2e-2, (k, float(per[k]), ref)
        else:
            close(per[k], g["loss/" + k], LOSS_TOL, 0, "loss/" + k)
    if bf:
        assert worst > 1e-7, "bf16 mode reproduced the fp32 losses exactly: the bf16 path did not run"
    else:
        close(total, g["loss/total"], LOSS_TOL, 0, "loss/total")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_g16b_cfg5_full_per_gpu_size_n224_backward(dev, lib, precision):
    """BASELINE configs[4] at its OWN per-GPU size, BACKWARD (VERDICT r5 "missing" 6): 224 rows x 576 regions x 80 tokens on the FULL model,
    one forward + backward of the four losses in ONE batch on the GPU against the reference's gradients (oracle/gen_golden_full.py g16b: the
    real reference run in chunks of 4 items and re-assembled by linearity of the mean losses, checked against g16's forward losses).
    fp32: every per-tensor gradient norm within 2e-4 relative, a 64-element strided slice of every gradient within 2e-4 of the tensor's
    largest slice entry + 1e-7, the set of tensors without a gradient equal.  bf16-resident: norms within 5 % (10 % for the co-attention
    query / key projections, see _bf16_check), slice cosines with the fp32 reference > 0.995 overall / > 0.93 per tensor, never bit-equal to fp32."""
    from ytvln import ops, synth
    g = gold("g16b_cfg5_full_n224_grads.npz")
    args = args_ns(**PRETRAIN)
    model, W = build_lily(dev, FULL_CFG, args, seed=34)
    batch = synth.to_torch(synth.make_batch(bs=32, K=7, T=80, frames=16, boxes=36, seed=44, ignore_rank_frac=0.0), dev)
    if precision == "bf16":
        _bf16_check(model, batch, args, g)
        return
    model.train()
    outputs, total, per = losses_of(model, batch, args)
    total.backward()
    for k in ("vision", "language", "ranking", "traj"):
        close(per[k], g["loss/" + k], LOSS_TOL, 0, "loss/" + k)
    pd = dict(model.named_parameters())
    assert {n for n, p in model.named_parameters() if p.grad is None} == set(g["unused"].tolist())
    bad = []
    for n, ref, sl in zip(g["grad_names"].tolist(), g["grad_norms"], g["grad_slices"]):
        gr = pd[n].grad.detach().reshape(-1)
        norm = float(gr.double().norm())
        if abs(norm - float(ref)) > 2e-4 * float(ref) + 1e-6:
            bad.append((n, "norm", norm, float(ref)))
        st = max(1, gr.numel() // 64)
        got = gr[::st][:64].float().cpu().numpy()
        err = float(np.abs(got - sl[: got.size]).max())
        if err > 2e-4 * float(np.abs(sl).max()) + 1e-7:
            bad.append((n, "slice", err, float(np.abs(sl).max())))
    assert not bad, bad[:10]


def test_g11_cfg2_full_size_n56_gradients(dev, lib):
    """BASELINE configs[1] at the size bench.py runs: bs = 8 items x K = 7 = 56 rows, T = 80, R = 288, FULL model -- losses, logit slices
    and checksums, ALL per-tensor gradient norms and the post-AdamW parameter summaries against the reference (round 1 checked gradients
    at N = 7 only)."""
    from ytvln import synth
    g = gold("g11_cfg2_full_n56.npz")
    args = args_ns(**PRETRAIN)
    model, W = build_lily(dev, FULL_CFG, args, seed=32)
    batch = synth.to_torch(synth.make_batch(bs=8, K=7, T=80, frames=8, boxes=36, seed=42), dev)
    check_summaries(model, W, batch, args, g, float(g["lr"]))      # (per-tensor bar inside: |norm - ref| <= 2e-4 ref + 1e-6)


def test_g12_cfg4_finetune_full_size_n96(dev, lib):
    """BASELINE configs[3] at its per-GPU size: train.py --ranking fine-tune, bs = 16 items x K = 6 = 96 rows, R = 7 x 36 = 252 (not a
    multiple of the 32-row attention tiles), ranking head only -- the 1e-4 bar on the FULL model."""
    from ytvln import synth
    g = gold("g12_cfg4_full_n96.npz")
    args = args_ns(ranking=True, pretrain=False, num_negatives=2)
    model, W = build_lily(dev, FULL_CFG, args, seed=33)
    batch = synth.to_torch(synth.make_batch(bs=16, K=6, T=80, frames=7, boxes=36, seed=43, finetune_heading=True), dev)
    check_summaries(model, W, batch, args, g, float(g["lr"]))


@pytest.mark.parametrize("case", ["g0", "g1", "g10", "g11", "g12"])
def test_fp32x3_meets_the_fp32_bar_on_every_golden(dev, lib, case):
    """the opt-in fp32x3 projections against EVERY model golden with the native path's own tolerances (g2 / g4 are
    covered by test_g2_g4_full_model_fp32x3_meets_the_fp32_bar, g3 below)."""
    from ytvln import ops, synth
    ops.set_matmul_precision("fp32x3")
    try:
        if case == "g0":
            test_g0_micro_everything(dev, lib)
        elif case == "g1":
            test_g1_tiny_masked_language(dev, lib)
        elif case == "g10":
            g = gold("g10_cfg5_long_n14.npz")
            args = args_ns(**PRETRAIN)
            model, W = build_lily(dev, FULL_CFG, args, seed=31)
            check_summaries(model, W, synth.to_torch(synth.make_batch(bs=2, K=7, T=80, frames=16, boxes=36, seed=41, ignore_rank_frac=0.0), dev),
                            args, g, float(g["lr"]))
        elif case == "g11":
            test_g11_cfg2_full_size_n56_gradients(dev, lib)
        else:
            test_g12_cfg4_finetune_full_size_n96(dev, lib)
    finally:
        ops.set_matmul_precision("fp32")


def test_fp32x3_multimodal_pretraining_golden(dev, lib):
    from ytvln import ops
    ops.set_matmul_precision("fp32x3")
    try:
        test_g3_multimodal_pretraining(dev, lib)
    finally:
        ops.set_matmul_precision("fp32")


def _bert_model(dev, seed, **over):
    from ytvln import synth
    from ytvln.vilbert import BertConfig, BertModel
    m = BertModel(BertConfig(**cfg_dict("tiny_2_2_1.json", **{**ZERO_DROP, **over})))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, seed).items()})
    return m.to(dev).train()


def _check_bert(m, inputs, g, tag):
    seq_t, seq_v, pool_t, pool_v, _ = m(*inputs)
    loss = (pool_t * pool_v).sum() + 0.01 * seq_t.sum() + 0.01 * seq_v.sum()
    loss.backward()
    for name, t in (("seq_t", seq_t), ("seq_v", seq_v), ("pool_t", pool_t), ("pool_v", pool_v)):
        close(t, g[f"{tag}/{name}"], 1e-4, 1e-4, f"{tag}/{name}")
    close(loss, g[tag + "/loss"], 2e-3, 1e-5, tag + "/loss")
    pd = dict(m.named_parameters())
    assert {n for n, p in pd.items() if p.grad is not None} == set(g[tag + "/grad_names"].tolist())
    for n, ref in zip(g[tag + "/grad_names"].tolist(), g[tag + "/grad_norms"]):
        got = float(pd[n].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * ref + 1e-5, (tag, n, got, float(ref))


def test_g13_in_batch_pairs_fast_mode_and_predict_feature(dev, lib):
    """The three switches that are off in every target config (vilbert.py:771-782, 1391, 1430-1434) against the reference's own outputs
    (oracle/gen_golden_branches.py): in_batch_pairs (3 texts x 3 images -> 9 rows), fast_mode (1 text against 4 images) and the
    predict_feature MSE loss -- outputs, the loss and every gradient norm."""
    from ytvln import synth
    from ytvln.vilbert import BertConfig, BertForMultiModalPreTraining
    g = gold("g13_branches.npz")

    def inputs(nb):
        b = synth.to_torch(nb, dev)
        return b[6][:, 0], b[1][:, 0], b[2][:, 0], b[10][:, 0], b[7][:, 0], b[3][:, 0]

    _check_bert(_bert_model(dev, 21, in_batch_pairs=True), inputs(synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=51)), g, "pairs")
    ids, feat, loc, seg, tmask, vmask = inputs(synth.make_batch(bs=4, K=1, T=12, frames=2, boxes=5, seed=52))
    _check_bert(_bert_model(dev, 22, fast_mode=True), (ids[:1], feat, loc, seg[:1], tmask[:1], vmask), g, "fast")

    cfg = BertConfig(**cfg_dict("tiny_2_2_1.json", **ZERO_DROP, predict_feature=True))
    m = BertForMultiModalPreTraining(cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, 23).items()})
    m.to(dev).eval()          # eval like the fixture (the NSP head's hard-wired Dropout(0.1) would bring torch's RNG in)
    b = synth.to_torch(synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=53), dev)
    l = m(b[6][:, 0], b[1][:, 0], b[2][:, 0], None, b[7][:, 0], b[3][:, 0], b[8][:, 0], b[5][:, 0, 1:],
          torch.from_numpy(g["mse/img_target"]).to(dev), torch.from_numpy(g["mse/nsl"]).to(dev))
    (l[0] + l[1] + l[2]).sum().backward()
    for got, ref in zip(l, g["mse/losses"]):
        assert abs(float(got) - float(ref)) <= LOSS_TOL, (float(got), float(ref))
    pd = dict(m.named_parameters())
    assert {n for n, p in pd.items() if p.grad is not None} == set(g["mse/grad_names"].tolist())
    for n, ref in zip(g["mse/grad_names"].tolist(), g["mse/grad_norms"]):
        assert abs(float(pd[n].grad.double().norm()) - ref) <= 2e-4 * ref + 1e-5, n


def test_g17_fixed_layers(dev, lib):
    """`fixed_t_layer` / `fixed_v_layer` (vilbert.py:742-764): the first layers of each stream run under no_grad.  Against the reference's
    own run (oracle/gen_golden_fixed.py) on a deepened tiny config (4 text / 3 image layers, co-attention after (t 2, v 1) and (t 3, v 2))
    with fixed_t_layer = 2, fixed_v_layer = 1: outputs, the loss, the exact SET of parameters that receive a gradient (the frozen layers
    and the embeddings below them get None) and every gradient norm; and the same model with the switches off."""
    from ytvln import synth
    g = gold("g17_fixed_layers.npz")
    deep = dict(num_hidden_layers=4, v_num_hidden_layers=3, t_biattention_id=[2, 3], v_biattention_id=[1, 2])
    b = synth.to_torch(synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=61), dev)
    inputs = (b[6][:, 0], b[1][:, 0], b[2][:, 0], b[10][:, 0], b[7][:, 0], b[3][:, 0])
    _check_bert(_bert_model(dev, 31, fixed_t_layer=2, fixed_v_layer=1, **deep), inputs, g, "fixed")
    _check_bert(_bert_model(dev, 31, **deep), inputs, g, "free")
    assert len(g["fixed/grad_names"]) < len(g["free/grad_names"])


TRAINMODE_RECIPE = dict(bs=2, K=3, T=16, frames=2, boxes=4, seed=31, ignore_rank_frac=0.0)


def _trainmode_run(dev, seed, steps, dropout, lr, total_steps, w_seed):
    """The recipe of oracle/gen_golden_trainmode.py on the HIP path: tiny config, fixed batch and initial weights, get_optimization's
    AdamW + WarmupLinear, `steps` optimizer steps; returns [steps, 5] losses (total, ranking, traj, vision, language) and the model."""
    from ytvln import ops, synth
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True, learning_rate=lr)
    ops.DropoutState.manual_seed(None)               # the mask stream follows torch's seed, like the reference's (utils/misc.py:37-45)
    torch.manual_seed(seed)
    over = {} if dropout else dict(ZERO_DROP)
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    cfg = BertConfig(**cfg_dict("tiny_2_2_1.json", **over))
    cfg.args = args
    model = Lily(cfg, dropout_prob=0.1 if dropout else 0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, w_seed).items()})
    model = model.to(dev).train()
    batch = synth.to_torch(synth.make_batch(**TRAINMODE_RECIPE), dev)
    opt, sched, _, _ = get_optimization(args, model, total_steps, None)
    torch.manual_seed(seed)
    rows = []
    for _ in range(steps):
        _, total, per = losses_of(model, batch, args)
        total.backward()
        rows.append(torch.stack([total.detach()] + [per[t].detach() for t in ("ranking", "traj", "vision", "language")]))
        opt.step(); sched.step(); opt.zero_grad()
    return torch.stack(rows).double().cpu().numpy(), model


@pytest.mark.gpu
def test_g15_finite_20_step_trajectory(dev, lib):
    """A FINITE multi-step loss trajectory from the real reference (g0's is NaN in the reference itself: its `opt_mask` hole): tiny config,
    all four heads, p = 0, 20 AdamW steps with the warm-up schedule -- every loss of every step, parameter norms after step 3 and 20."""
    g = gold("g15_tiny_traj20.npz")
    steps = int(g["steps"])
    rows, model = _trainmode_run(dev, 0, steps, False, float(g["lr"]), int(g["total_steps"]), int(g["w_seed"]))
    ref = g["losses"]
    assert np.isfinite(ref).all() and ref.shape == rows.shape
    for s in range(steps):      # fp32 rounding differences compound slowly along the trajectory: the bar widens with the step
        tol = LOSS_TOL if s < 3 else LOSS_TOL * (1 + s)
        assert np.abs(rows[s] - ref[s]).max() <= tol + 2e-5 * np.abs(ref[s]).max(), (s, rows[s], ref[s])
    pd = dict(model.named_parameters())
    for n, nr, sr in zip(g["post_names"].tolist(), g["post_norm"], g["post_sum"]):
        p = pd[n].detach().double()
        # (20 Adam steps of 1e-3: an element whose gradient is rounding noise may walk the other way -- absolute room of 2e-4 per sqrt(element))
        assert abs(float(p.norm()) - nr) <= 2e-4 * nr + 2e-4 * p.numel() ** 0.5, f"norm after {steps} steps: {n}"
    _, model3 = _trainmode_run(dev, 0, 3, False, float(g["lr"]), int(g["total_steps"]), int(g["w_seed"]))
    pd = dict(model3.named_parameters())
    for n, nr, sr in zip(g["post_names"].tolist(), g["post3_norm"], g["post3_sum"]):
        p = pd[n].detach().double()
        assert abs(float(p.norm()) - nr) <= 5e-6 * nr + 2e-6 * p.numel() ** 0.5, f"norm after 3 steps: {n}"
        assert abs(float(p.sum()) - sr) <= 1e-5 * float(p.abs().sum()) + 2e-6 * p.numel() ** 0.5, f"sum after 3 steps: {n}"


@pytest.mark.gpu
def test_g14_train_mode_matches_the_reference_in_distribution(dev, lib):
    """SURVEY H1 / VERDICT r2: with dropout ON the masks differ by construction (the reference draws from torch's Philox stream), so parity is
    statistical: over 32 mask seeds the mean of every loss at every one of 20 optimizer steps must sit within 4 sigma_ref / sqrt(32) of the
    reference's mean over its 128 seeds (a wrong keep-scale, a mask applied twice or a missing dropout site shifts it by tens of sigma), and
    the spread over seeds must be of the reference's size."""
    g = gold("g14_trainmode_stats.npz")
    steps, n_hip = int(g["steps"]), 32
    runs = np.stack([_trainmode_run(dev, 1000 + s, steps, True, float(g["lr"]), int(g["total_steps"]), int(g["w_seed"]))[0] for s in range(n_hip)])
    assert np.isfinite(runs).all()
    mean, std = runs.mean(0), runs.std(0, ddof=1)
    ref_mean, ref_std = g["mean"], g["std"]
    n_ref = int(g["n_seeds"])
    bound = 4.0 * ref_std * np.sqrt(1.0 / n_hip + 1.0 / n_ref) + 1e-3
    worst = np.abs(mean - ref_mean) / (ref_std * np.sqrt(1.0 / n_hip + 1.0 / n_ref) + 1e-9)
    assert (np.abs(mean - ref_mean) <= bound).all(), f"largest deviation {worst.max():.2f} sigma at (step, loss) {np.unravel_index(worst.argmax(), worst.shape)}"
    ratio = std[:, 0] / ref_std[:, 0]
    assert (ratio > 0.5).all() and (ratio < 2.0).all(), ratio
    # and the eval-mode trajectory is NOT inside that band: the test can tell dropout from no dropout
    det = gold("g15_tiny_traj20.npz")["losses"]
    assert (np.abs(det[5:, 0] - ref_mean[5:, 0]) > bound[5:, 0]).any()
