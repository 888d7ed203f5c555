"""CPU: the oracle (oracle/vilbert_ref.py) replayed against the fixtures generated from the imported reference.

This is what pins the oracle: the reference has no tests / golden vectors of its own, so
tests/golden/*.npz were produced by running the real reference in the build container (oracle/gen_golden.py).
"""
import numpy as np
import pytest
import torch

import vilbert_ref as O
from helpers import ZERO_DROP, cfg_dict, close, gold, rel_l2
from ytvln import synth


def flags(**kw):
    return O.TaskFlags(**kw)


def state(W):
    return {k: torch.from_numpy(v).clone() for k, v in W.items()}


def test_g5_unit_kats():
    k = gold("g5_kats.npz")
    t = lambda n: torch.from_numpy(k[n])   # noqa: E731
    close(O.layer_norm(t("ln/x"), t("ln/w"), t("ln/b")), k["ln/y"], 1e-6, 1e-6, "layer_norm")
    close(O._act("gelu", t("gelu/x")), k["gelu/y"], 1e-7, 1e-7, "gelu")
    add = (1.0 - t("softmax/mask"))[:, None, None, :] * -10000.0
    close(torch.softmax(t("softmax/s") / 8 ** 0.5 + add, -1), k["softmax/p"], 1e-7, 1e-6, "masked softmax")
    assert np.array_equal(O.pad_packed(t("pad_packed/t"), t("pad_packed/mask")).numpy(), k["pad_packed/out"])
    st = O.AdamWState()
    p = {"w": t("adamw/p0").clone()}
    for i in range(3):
        O.adamw_step(p, {"w": torch.from_numpy(k["adamw/grads"][i])}, st, 1e-2, 0.01)
        close(p["w"], k[f"adamw/p{i + 1}"], 1e-7, 1e-6, f"adamw step {i + 1}")
    close(st.exp_avg["w"], k["adamw/m"], 1e-9, 1e-6)
    close(st.exp_avg_sq["w"], k["adamw/v"], 1e-10, 1e-6)
    warm, tot = k["sched/warm_total"]
    for s, lam in zip(k["sched/steps"], k["sched/lambda"]):
        assert abs(O.warmup_linear(int(s), warm, tot) - lam) < 1e-15
    assert (warm, tot) == O.schedule_totals(50, 1, 2)


def test_g0_micro_forward_losses_grads_and_adamw():
    g = gold("g0_micro.npz")
    cfg = O.RefConfig(**cfg_dict("micro.json", **ZERO_DROP))
    fl = flags(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    W = {k[2:]: g[k] for k in g.files if k.startswith("w/")}
    batch = [torch.from_numpy(g["in_%02d" % i]) for i in range(16)]
    col = {}
    ids, feat, loc, seg, imask, vmask = O.model_input(batch)
    with torch.no_grad():
        out = O.lily_forward(state(W), cfg, fl, ids, feat, loc, seg, imask, vmask, collect=col)
        total, per = O.total_loss(batch, out, fl)
    close(col["embedding_output"], g["embedding_output"], 1e-5, 1e-5)
    close(col["v_embedding_output"], g["v_embedding_output"], 1e-5, 1e-5)
    for name in ("t0", "t1"):
        close(col[name + ".t"], g[name + ".t"], 2e-5, 2e-5, name)
        close(col[name + ".probs"], g[name + ".probs"], 1e-6, 1e-5, name)
    for name in ("v0", "v1"):
        close(col[name + ".v"], g[name + ".v"], 2e-5, 2e-5, name)
    close(col["c0.v"], g["c0.v"], 2e-5, 2e-5)
    close(col["c0.t"], g["c0.t"], 2e-5, 2e-5)
    close(col["c0.probs"][0], g["c0.probs1"], 1e-6, 1e-5)
    close(col["c0.probs"][1], g["c0.probs2"], 1e-6, 1e-5)
    for k in ("ranking", "traj", "vision", "language"):
        close(out[k], g["logits/" + k], 2e-5, 2e-5, k)
        close(per[k], g["loss/" + k], 1e-6, 1e-6, k)
    close(total, g["loss/total"], 1e-6, 1e-6)
    # 3 optimizer steps
    S = state(W)
    st = O.AdamWState()
    warm, tot = O.schedule_totals(10, 1, 1)
    for step in range(3):
        lr = 1e-3 * O.warmup_linear(step, warm, tot)
        assert abs(lr - float(g[f"step{step}.lr"])) < 1e-12
        loss, _, grads, _ = O.train_step(S, cfg, fl, batch, st, lr)
        close(loss, g[f"step{step}.loss"], 1e-6, 1e-6)
        if step == 0:
            unused = {n for n, v in grads.items() if v is None}
            assert unused == set(g["unused"].tolist())
            for n, v in grads.items():
                if v is not None:
                    assert rel_l2(v, g["grad/" + n]) < 1e-4 or float(np.linalg.norm(g["grad/" + n])) < 1e-7, n
    for n in S:
        if ("after3/" + n) in g.files:
            close(S[n], g["after3/" + n], 1e-7, 1e-6, n)


def _summaries(cfgname, fl, seed, nb, g):
    cfg = O.RefConfig(**cfg_dict(cfgname, **ZERO_DROP))
    import json, os
    from conftest import GOLD
    shapes = json.load(open(os.path.join(GOLD, "state_dict_schema.json")))["Lily/" + cfgname]["shapes"]
    W = synth.make_weights({k: tuple(v) for k, v in shapes.items()}, seed)
    S = state(W)
    batch = synth.to_torch(nb)
    loss, per, grads, out = O.train_step(S, cfg, fl, batch, O.AdamWState(), float(g["lr"]))
    close(loss, g["loss/total"], 2e-6, 2e-6)
    for k, v in per.items():
        close(v, g["loss/" + k], 2e-6, 2e-6, k)
    for k, v in out.items():
        stride = int(g["logits_stride/" + k])
        ref = g["logits/" + k]
        flat = v.detach().reshape(v.shape[0], -1)
        got = v.detach() if ref.shape == tuple(v.shape) else flat[:, ::stride][:, :ref.shape[1]]
        close(got, ref, 5e-5, 5e-5, k)
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        assert abs(float(grads[n].double().norm()) - ref) <= 2e-4 * ref + 1e-7, n
    assert {n for n, v in grads.items() if v is None} == set(g["unused"].tolist())
    for n, s_ref, n_ref in zip(g["param_names"].tolist(), g["post_sum"], g["post_norm"]):
        assert abs(float(S[n].double().norm()) - n_ref) <= 2e-6 * n_ref + 1e-7, n


def test_g1_tiny_mlm():
    _summaries("tiny_2_2_1.json", flags(masked_language=True), 12, synth.make_batch(bs=2, K=7, T=16, frames=1, boxes=8, seed=22),
               gold("g1_tiny_mlm.npz"))


def test_g2_full_all_losses():
    _summaries("bert_base_6_layer_6_connect.json", flags(ranking=True, traj_judge=True, masked_vision=True, masked_language=True), 13,
               synth.make_batch(bs=1, K=7, T=80, frames=8, boxes=36, seed=23, ignore_rank_frac=0.0), gold("g2_full_n7.npz"))


def test_g3_multimodal_pretraining():
    g = gold("g3_multimodal_pretraining.npz")
    import json, os
    from conftest import GOLD
    cfg = O.RefConfig(**cfg_dict("tiny_2_2_1.json", **ZERO_DROP))
    shapes = json.load(open(os.path.join(GOLD, "state_dict_schema.json")))["BertForMultiModalPreTraining/tiny_2_2_1.json"]["shapes"]
    S = state(synth.make_weights({k: tuple(v) for k, v in shapes.items()}, 14))
    b = synth.to_torch(synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=24))
    ids, feat, loc, vmask = b[6][:, 0], b[1][:, 0], b[2][:, 0], b[3][:, 0]
    with torch.no_grad():
        l = O.multimodal_pretraining_forward(S, cfg, ids, feat, loc, None, b[7][:, 0], vmask, b[8][:, 0], b[5][:, 0, 1:], b[4][:, 0, 1:],
                                             torch.from_numpy(g["nsl"]))
    for i, n in enumerate(("masked_lm_loss", "masked_img_loss", "next_sentence_loss")):
        close(l[i], g[n], 1e-6, 1e-6, n)


def test_encoder_schedule_matches_survey():
    full = O.RefConfig(**cfg_dict("bert_base_6_layer_6_connect.json"))
    order = "".join(f"{k.upper()}{i}," for k, i in O.encoder_schedule(full))
    assert order == "T0,T1,T2,T3,T4,T5,C0,V0,T6,C1,V1,T7,C2,V2,T8,C3,V3,T9,C4,V4,T10,C5,V5,T11,"
    tiny = O.RefConfig(**cfg_dict("tiny_2_2_1.json"))
    assert "".join(f"{k.upper()}{i}," for k, i in O.encoder_schedule(tiny)) == "V0,T0,C0,V1,T1,"


def test_g7_masking_restatement_matches_reference():
    """oracle.randomize_tokens / randomize_regions against the outputs of the reference's functions (utils/dataset/common.py:213-300)
    on the same inputs and the same uniform draws (oracle/gen_golden_masking.py)."""
    g = gold("g7_masking.npz")
    t = lambda k: torch.from_numpy(g[k])   # noqa: E731
    tok, tgt = O.randomize_tokens(t("tokens"), t("mask"), t("p_tok"), t("rnd_tok"))
    assert torch.equal(tok, t("out_tok")) and torch.equal(tgt, t("tgt_tok"))
    f, tg, m = O.randomize_regions(t("feats"), t("probs"), t("rmask"), t("p_reg"))
    assert torch.equal(f, t("out_f")) and torch.equal(tg, t("out_t")) and torch.equal(m, t("out_m"))
    # the fixture exercises every branch
    p = t("p_tok") * t("mask").float()
    assert int(((p >= 0.85) & (p < 0.97)).sum()) > 0 and int(((p >= 0.97) & (p < 0.985)).sum()) > 0 and int((p >= 0.985).sum()) > 0


def test_g15_oracle_follows_the_reference_trajectory():
    """The finite multi-step golden (oracle/gen_golden_trainmode.py g15): the oracle's first six AdamW steps on the tiny config reproduce the
    reference's loss trajectory, and g14's seed statistics are self-consistent (finite, mean / std recomputable from the stored runs)."""
    g = gold("g15_tiny_traj20.npz")
    cfg = O.RefConfig(**cfg_dict("tiny_2_2_1.json", **ZERO_DROP))
    fl = flags(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    import json, os
    from conftest import GOLD
    shapes = json.load(open(os.path.join(GOLD, "state_dict_schema.json")))["Lily/tiny_2_2_1.json"]["shapes"]
    W = synth.make_weights({k: tuple(v) for k, v in shapes.items()}, int(g["w_seed"]))
    batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=16, frames=2, boxes=4, seed=31, ignore_rank_frac=0.0))
    S, st = state(W), O.AdamWState()
    warm, tot = O.schedule_totals(int(g["total_steps"]), 1, 1)
    assert np.isfinite(g["losses"]).all()
    for step in range(6):
        loss, _, _, _ = O.train_step(S, cfg, fl, batch, st, float(g["lr"]) * O.warmup_linear(step, warm, tot))
        assert abs(float(loss) - g["losses"][step, 0]) <= 2e-5 * (step + 1), (step, float(loss), g["losses"][step, 0])
    s = gold("g14_trainmode_stats.npz")
    runs = s["losses"].astype(np.float64)
    assert runs.shape == (int(s["n_seeds"]), int(s["steps"]), 5) and np.isfinite(runs).all()
    assert np.allclose(runs.mean(0), s["mean"], atol=1e-5) and np.allclose(runs.std(0, ddof=1), s["std"], atol=1e-5)
    assert (s["std"][:, 0] > 0).all()          # dropout really was on: the seeds differ


def test_g16b_gradient_fixture_is_the_backward_of_g16():
    """The chunked reference backward at configs[4]'s own size (oracle/gen_golden_full.py g16b) re-assembles g16's forward losses, names
    every parameter of the full model exactly once (gradient or `unused`), and its unused set is the reference's H3/H5 set of g10 / g11."""
    f, b = gold("g16_cfg5_full_n224.npz"), gold("g16b_cfg5_full_n224_grads.npz")
    for k in ("vision", "language", "ranking", "traj"):
        assert abs(float(b["loss/" + k]) - float(f["loss/" + k])) <= 2e-5 * max(1.0, abs(float(f["loss/" + k])))
    g10 = gold("g10_cfg5_long_n14.npz")
    assert sorted(b["unused"].tolist()) == sorted(g10["unused"].tolist())
    assert sorted(b["grad_names"].tolist()) == sorted(g10["grad_names"].tolist())
    assert b["grad_slices"].shape == (len(b["grad_names"]), 64) and np.isfinite(b["grad_norms"]).all() and (b["grad_norms"] > 0).all()
