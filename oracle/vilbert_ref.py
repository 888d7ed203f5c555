"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A functional, pure-torch (CPU, fp32/fp64) restatement of the one hot path of
JeremyLinky/YouTube-VLN that this repository accelerates: the ViLBERT two-stream
forward, the four pre-training losses, the HF-1.2-style AdamW step and the LR schedules.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Every function cites the reference lines it restates (paths relative to the reference
checkout).  The state is a plain ``dict[str, Tensor]`` keyed exactly like the reference
``Lily`` / ``BertForMultiModalPreTraining`` ``state_dict()`` so the same numpy-seeded weights
can be loaded into the reference, into this oracle and into the HIP product.

Pinning: the reference ships no tests or golden vectors of its own, so the
oracle is pinned by fixtures generated *from the imported reference itself* in the build
container (`oracle/gen_golden.py` -> `tests/golden/*.npz`); `tests/test_oracle_golden.py` replays
them on every run.

Dropout: parity is defined for eval mode / p = 0 (the reference draws its masks from torch's global
Philox stream, which no independent implementation can reproduce).  ``drop`` > 0 is supported
only so the CPU baseline timing does the same amount of work as the reference's train mode.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# configuration (vilbert/vilbert.py:129-175 -- field names and defaults)
# --------------------------------------------------------------------------------------
@dataclass
class RefConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "gelu"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    initializer_range: float = 0.02
    v_feature_size: int = 2048
    v_target_size: int = 1601
    v_hidden_size: int = 768
    v_num_hidden_layers: int = 3
    v_num_attention_heads: int = 12
    v_intermediate_size: int = 3072
    bi_hidden_size: int = 1024
    bi_num_attention_heads: int = 16
    v_attention_probs_dropout_prob: float = 0.1
    v_hidden_act: str = "gelu"
    v_hidden_dropout_prob: float = 0.1
    v_initializer_range: float = 0.2
    v_biattention_id: Sequence[int] = (0, 1)
    t_biattention_id: Sequence[int] = (10, 11)
    order_hidden_size: int = 512
    predict_feature: int = False
    fast_mode: int = False
    fixed_v_layer: int = 0
    fixed_t_layer: int = 0
    in_batch_pairs: int = False
    fusion_method: str = "mul"
    intra_gate: int = False
    with_coattention: int = True
    ranking: bool = True
    masked_language: bool = False
    masked_vision: bool = False

    def __post_init__(self):  # vilbert.py:172-175
        assert len(self.v_biattention_id) == len(self.t_biattention_id)
        assert max(self.v_biattention_id) < self.v_num_hidden_layers
        assert max(self.t_biattention_id) < self.num_hidden_layers


def _act(name: str, x: Tensor) -> Tensor:
    """vilbert.py:113-126 (ACT2FN). gelu is the exact erf form of :119."""
    if name == "gelu":
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "relu":
        return torch.relu(x)
    if name == "swish":
        return x * torch.sigmoid(x)
    raise KeyError(name)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-12) -> Tensor:
    """vilbert.py:213-217 -- biased variance, eps inside the sqrt."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def _lin(S: State, name: str, x: Tensor) -> Tensor:
    return F.linear(x, S[name + ".weight"], S.get(name + ".bias"))


def _ln(S: State, name: str, x: Tensor) -> Tensor:
    return layer_norm(x, S[name + ".weight"], S[name + ".bias"])


def _drop(x: Tensor, p: float) -> Tensor:
    return F.dropout(x, p, True) if p > 0.0 else x


# --------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------
def text_embeddings(S: State, ids: Tensor, type_ids: Optional[Tensor], p: float = 0.0,
                    pre: str = "bert.embeddings.") -> Tensor:
    """vilbert.py:240-256: LN(word[ids] + pos[arange(T)] + type[tt]) -> dropout."""
    T = ids.size(1)
    pos = torch.arange(T, dtype=torch.long, device=ids.device).unsqueeze(0).expand_as(ids)
    if type_ids is None:
        type_ids = torch.zeros_like(ids)
    e = (S[pre + "word_embeddings.weight"][ids]
         + S[pre + "position_embeddings.weight"][pos]
         + S[pre + "token_type_embeddings.weight"][type_ids])
    return _drop(_ln(S, pre + "LayerNorm", e), p)


def image_embeddings(S: State, feat: Tensor, loc: Tensor, p: float = 0.0,
                     pre: str = "bert.v_embeddings.") -> Tensor:
    """vilbert.py:1356-1370: LN(W_f feat + W5 loc[:5] + W4 loc[5:9] + W2 loc[9:11] + E32[loc[11]])."""
    img = _lin(S, pre + "image_embeddings", feat)
    a = _lin(S, pre + "image_location_embeddings", loc[..., :5])
    b = _lin(S, pre + "image_orientation_embeddings", loc[..., 5:9])
    c = _lin(S, pre + "image_next_orientation_embeddings", loc[..., 9:11])
    d = S[pre + "image_sequence_embeddings.weight"][loc[..., 11].long()]
    return _drop(_ln(S, pre + "LayerNorm", img + (a + b + c + d)), p)


# --------------------------------------------------------------------------------------
# attention building blocks
# --------------------------------------------------------------------------------------
def _heads(x: Tensor, h: int) -> Tensor:
    """transpose_for_scores, vilbert.py:276-282."""
    n, t, hd = x.shape
    return x.view(n, t, h, hd // h).permute(0, 2, 1, 3)


def _attend(q: Tensor, k: Tensor, v: Tensor, add_mask: Tensor, h: int, p: float):
    """vilbert.py:289-311 (and :418-440, :577-616): softmax(QK^T/sqrt(d) + mask) V, heads merged."""
    q, k, v = _heads(q, h), _heads(k, h), _heads(v, h)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.size(-1)) + add_mask
    pr = torch.softmax(s, dim=-1)
    ctx = torch.matmul(_drop(pr, p), v).permute(0, 2, 1, 3).contiguous()
    return ctx.view(ctx.size(0), ctx.size(1), -1), pr


def self_layer(S: State, pre: str, x: Tensor, add_mask: Tensor, heads: int, act: str,
               p_attn: float, p_hid: float):
    """BertLayer / BertImageLayer, vilbert.py:371-382 / :498-509 (identical structure)."""
    a = pre + "attention."
    ctx, probs = _attend(_lin(S, a + "self.query", x), _lin(S, a + "self.key", x),
                         _lin(S, a + "self.value", x), add_mask, heads, p_attn)
    att = _ln(S, a + "output.LayerNorm", _drop(_lin(S, a + "output.dense", ctx), p_hid) + x)  # :321-325
    inter = _act(act, _lin(S, pre + "intermediate.dense", att))                                # :351-354
    out = _ln(S, pre + "output.LayerNorm", _drop(_lin(S, pre + "output.dense", inter), p_hid) + att)  # :364-368
    return out, probs


def connection_layer(S: State, pre: str, cfg: RefConfig, v: Tensor, v_mask: Tensor, t: Tensor,
                     t_mask: Tensor, drop: bool):
    """BertConnectionLayer, vilbert.py:652-679, with BertBiAttention :552-618 and BertBiOutput :638-650.

    stream 1 = vision, stream 2 = text.  ctx1 = text queries over image keys/values,
    ctx2 = image queries over text keys/values; biOutput receives them swapped (:671).
    """
    h = cfg.bi_num_attention_heads
    pv = cfg.v_attention_probs_dropout_prob if drop else 0.0
    pt = cfg.attention_probs_dropout_prob if drop else 0.0
    phv = cfg.v_hidden_dropout_prob if drop else 0.0
    pht = cfg.hidden_dropout_prob if drop else 0.0
    b = pre + "biattention."
    q1, k1, v1 = _lin(S, b + "query1", v), _lin(S, b + "key1", v), _lin(S, b + "value1", v)
    q2, k2, v2 = _lin(S, b + "query2", t), _lin(S, b + "key2", t), _lin(S, b + "value2", t)
    ctx1, p1 = _attend(q2, k1, v1, v_mask, h, pv)   # :577-594  [N,T,Hb]
    ctx2, p2 = _attend(q1, k2, v2, t_mask, h, pt)   # :597-616  [N,R,Hb]
    o = pre + "biOutput."
    av = _ln(S, o + "LayerNorm1", _drop(_lin(S, o + "dense1", ctx2), phv) + v)   # :641-647
    at = _ln(S, o + "LayerNorm2", _drop(_lin(S, o + "dense2", ctx1), pht) + t)   # :644-648
    iv = _act(cfg.v_hidden_act, _lin(S, pre + "v_intermediate.dense", av))
    ov = _ln(S, pre + "v_output.LayerNorm", _drop(_lin(S, pre + "v_output.dense", iv), phv) + av)
    it = _act(cfg.hidden_act, _lin(S, pre + "t_intermediate.dense", at))
    ot = _ln(S, pre + "t_output.LayerNorm", _drop(_lin(S, pre + "t_output.dense", it), pht) + at)
    return ov, ot, (p1, p2)


def encoder_schedule(cfg: RefConfig) -> List[Tuple[str, int]]:
    """Order in which BertEncoder.forward (vilbert.py:737-811) visits its layers."""
    order: List[Tuple[str, int]] = []
    vs = ts = 0
    for c, (ve, te) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        order += [("v", i) for i in range(vs, ve)]
        order += [("t", i) for i in range(ts, te)]
        if cfg.with_coattention:
            order.append(("c", c))
        vs, ts = ve, te
    order += [("v", i) for i in range(vs, cfg.v_num_hidden_layers)]
    order += [("t", i) for i in range(ts, cfg.num_hidden_layers)]
    return order


def bert_model(S: State, cfg: RefConfig, ids: Tensor, feat: Tensor, loc: Tensor,
               type_ids: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
               image_attention_mask: Optional[Tensor] = None, drop: bool = False,
               collect: Optional[dict] = None, pre: str = "bert."):
    """BertModel.forward, vilbert.py:1242-1337 (fixed_*_layer / in_batch_pairs / fast_mode off)."""
    assert not (cfg.fixed_t_layer or cfg.fixed_v_layer or cfg.in_batch_pairs or cfg.fast_mode)
    if attention_mask is None:
        attention_mask = torch.ones_like(ids)
    if image_attention_mask is None:
        image_attention_mask = torch.ones(feat.size(0), feat.size(1)).type_as(ids)
    dt = S[pre + "embeddings.word_embeddings.weight"].dtype
    t_mask = (1.0 - attention_mask[:, None, None, :].to(dt)) * -10000.0        # :1268-1282
    v_mask = (1.0 - image_attention_mask[:, None, None, :].to(dt)) * -10000.0  # :1269-1287
    ph = cfg.hidden_dropout_prob if drop else 0.0
    t = text_embeddings(S, ids, type_ids, ph, pre + "embeddings.")
    v = image_embeddings(S, feat.to(dt), loc.to(dt), ph, pre + "v_embeddings.")   # dropout uses hidden_dropout_prob (:1354)
    if collect is not None:
        collect["embedding_output"], collect["v_embedding_output"] = t, v
    for kind, i in encoder_schedule(cfg):
        if kind == "t":
            t, pr = self_layer(S, f"{pre}encoder.layer.{i}.", t, t_mask, cfg.num_attention_heads, cfg.hidden_act,
                               cfg.attention_probs_dropout_prob if drop else 0.0, ph)
        elif kind == "v":
            v, pr = self_layer(S, f"{pre}encoder.v_layer.{i}.", v, v_mask, cfg.v_num_attention_heads,
                               cfg.v_hidden_act, cfg.v_attention_probs_dropout_prob if drop else 0.0,
                               cfg.v_hidden_dropout_prob if drop else 0.0)
        else:
            v, t, pr = connection_layer(S, f"{pre}encoder.c_layer.{i}.", cfg, v, v_mask, t, t_mask, drop)
        if collect is not None:
            if kind in ("t", "c"):
                collect[f"{kind}{i}.t"] = t
            if kind in ("v", "c"):
                collect[f"{kind}{i}.v"] = v
            collect[f"{kind}{i}.probs"] = pr
    pooled_t = torch.relu(_lin(S, pre + "t_pooler.dense", t[:, 0]))   # :827-833
    pooled_v = torch.relu(_lin(S, pre + "v_pooler.dense", v[:, 0]))   # :842-848
    return t, v, pooled_t, pooled_v


def pretraining_heads(S: State, cfg: RefConfig, seq_t: Tensor, seq_v: Tensor, pooled_t: Tensor,
                      pooled_v: Tensor, pre: str = "cls."):
    """BertPreTrainingHeads.forward, vilbert.py:939-954 (NSP-branch dropout omitted: p=0 parity)."""
    pooled = pooled_t * pooled_v if cfg.fusion_method == "mul" else pooled_t + pooled_v
    h = _ln(S, pre + "predictions.transform.LayerNorm",
            _act(cfg.hidden_act, _lin(S, pre + "predictions.transform.dense", seq_t)))       # :863-867
    # decoder weight is tied to the word embedding (:901); the key exists in the state dict as well.
    scores_t = F.linear(h, S[pre + "predictions.decoder.weight"]) + S[pre + "predictions.bias"]  # :904-907
    rel = _lin(S, pre + "bi_seq_relationship", pooled)
    # BertImgPredictionHeadTransform tests config.hidden_act but applies ACT2FN[hidden_act] (:874-879)
    hv = _ln(S, pre + "imagePredictions.transform.LayerNorm",
             _act(cfg.hidden_act, _lin(S, pre + "imagePredictions.transform.dense", seq_v)))
    scores_v = _lin(S, pre + "imagePredictions.decoder", hv)                                  # :966-969
    return scores_t, scores_v, rel


@dataclass
class TaskFlags:
    """The args fields Lily and the loop read (utils/cli.py; lily.py:117-127; utils_init.py:147-158)."""
    ranking: bool = False
    traj_judge: bool = False
    masked_vision: bool = False
    masked_language: bool = False
    pretrain: bool = True
    num_negatives: int = 2
    traj_loss_scale: float = 1.0
    not_traj_judge_data: bool = False
    model_name: str = "vilbert"


def lily_forward(S: State, cfg: RefConfig, flags: TaskFlags, ids, feat, loc, type_ids=None,
                 attention_mask=None, image_attention_mask=None, drop: bool = False,
                 lily_dropout: float = 0.1, collect: Optional[dict] = None) -> Dict[str, Tensor]:
    """Lily.forward, lily.py:58-129."""
    t, v, pt, pv = bert_model(S, cfg, ids, feat, loc, type_ids, attention_mask, image_attention_mask, drop, collect)
    lang, vis, _ = pretraining_heads(S, cfg, t, v, pt, pv)
    pooled = pt * pv if cfg.fusion_method == "mul" else pt + pv   # lily.py:93-98
    pooled = _drop(pooled, lily_dropout if drop else 0.0)          # lily.py:100
    if collect is not None:
        collect.update(sequence_output_t=t, sequence_output_v=v, pooled_output_t=pt, pooled_output_v=pv)
    out: Dict[str, Tensor] = {}
    if flags.ranking:
        out["ranking"] = _lin(S, "vil_logit", pooled)
    if flags.traj_judge:
        out["traj"] = _lin(S, "judge", pooled)
    if flags.masked_vision:
        out["vision"] = vis
    if flags.masked_language:
        out["language"] = lang
    return out


def multimodal_pretraining_forward(S: State, cfg: RefConfig, ids, feat, loc, type_ids=None, attention_mask=None,
                                   image_attention_mask=None, masked_lm_labels=None, image_label=None,
                                   image_target=None, next_sentence_label=None):
    """BertForMultiModalPreTraining.forward, vilbert.py:1396-1455 (predict_feature = False branch)."""
    t, v, pt, pv = bert_model(S, cfg, ids, feat, loc, type_ids, attention_mask, image_attention_mask)
    st, sv, rel = pretraining_heads(S, cfg, t, v, pt, pv)
    if masked_lm_labels is not None and next_sentence_label is not None and image_target is not None:
        sv = sv[:, 1:]                                                              # :1429
        kl = F.kl_div(F.log_softmax(sv, dim=2), image_target, reduction="none")     # :1437-1439
        m = (image_label == 1)
        img_loss = torch.sum(kl * m.unsqueeze(2).to(kl.dtype)) / max(torch.sum(m), 0)   # :1440-1442 (sic: max(.,0))
        lm_loss = F.cross_entropy(st.reshape(-1, cfg.vocab_size), masked_lm_labels.view(-1), ignore_index=-1)
        nsp = F.cross_entropy(rel.view(-1, 2), next_sentence_label.view(-1), ignore_index=-1)
        return lm_loss.unsqueeze(0), img_loss.unsqueeze(0), nsp.unsqueeze(0)
    return st, sv, rel


# --------------------------------------------------------------------------------------
# batch parsing and losses (utils/utils_init.py)
# --------------------------------------------------------------------------------------
def pad_packed(t: Tensor, mask: Tensor) -> Tensor:
    """utils/dataset/common.py:21-26."""
    mask = mask.bool()
    out = mask.clone().to(t.dtype)
    out[mask] = t
    out[~mask] = -float("inf")
    return out


def model_input(batch: Sequence[Tensor]):
    """get_model_input, utils/utils_init.py:34-77: (ids, feat, loc, segment_ids, instr_mask, image_mask)."""
    opt = batch[13]
    return batch[6][opt], batch[1][opt], batch[2][opt], batch[10][opt], batch[7][opt], batch[3][opt]


def task_loss(batch: Sequence[Tensor], outputs: Dict[str, Tensor], task: str, flags: TaskFlags,
              training: bool = True):
    """get_loss_correct, utils/utils_init.py:108-164.  Returns (loss, correct)."""
    opt = batch[13]
    dev = opt.device
    correct = torch.tensor(0.0, device=dev)
    if task == "vision":                                              # :117-128
        pred = outputs["vision"]
        pred = pred.reshape(-1, pred.shape[2])
        target = batch[4][opt].flatten(0, 1)
        tmask = batch[5][opt].flatten()
        loss = F.kl_div(F.log_softmax(pred, dim=-1), target.to(pred.dtype), reduction="none")
        loss = loss * tmask.unsqueeze(-1).to(pred.dtype)
        loss = torch.sum(loss) / max(1, int(torch.sum(tmask).item()))
    elif task == "language":                                          # :129-135
        pred = outputs["language"]
        loss = F.cross_entropy(pred.reshape(-1, pred.shape[-1]), batch[8][opt].flatten(), ignore_index=-1)
    elif task == "ranking":                                           # :136-146
        target = batch[0]
        pred = pad_packed(outputs["ranking"].squeeze(1), opt)
        if training:
            loss = F.cross_entropy(pred, target, ignore_index=-1)
            correct = torch.sum(torch.argmax(pred, 1) == target).float()
        else:
            loss = F.binary_cross_entropy_with_logits(pred, target.to(pred.dtype))
            correct = torch.sum(target.gather(1, torch.argmax(pred, 1).view(-1, 1))).float()
    elif task == "traj":                                              # :147-161
        pred = pad_packed(outputs["traj"].squeeze(1), opt)
        target = torch.zeros(pred.shape, device=dev).bool()
        if not (flags.ranking or flags.not_traj_judge_data):
            target[:, 0] = 1
        elif flags.pretrain:
            target[:, : (1 + flags.num_negatives)] = 1
        else:
            target[:, : -flags.num_negatives] = 1
        pos_weight = torch.tensor([target.shape[1] / target[0].sum() - 1], device=dev, dtype=pred.dtype)
        loss = F.binary_cross_entropy_with_logits(pred, target.to(pred.dtype), pos_weight=pos_weight)
        correct = torch.sum((pred.sigmoid() > 0.5) == target).float() / target.shape[1]
    else:
        raise KeyError(task)
    return loss, correct


TASK_ORDER = (("vision", "masked_vision"), ("language", "masked_language"), ("ranking", "ranking"), ("traj", "traj_judge"))


def total_loss(batch, outputs, flags: TaskFlags, training: bool = True):
    """Loss composition of train_epoch, utils/utils_init.py:217-224 (order vision, language, ranking, traj)."""
    total = None
    per_task: Dict[str, Tensor] = {}
    for task, flag in TASK_ORDER:
        if getattr(flags, flag):
            l, _ = task_loss(batch, outputs, task, flags, training)
            per_task[task] = l
            l = flags.traj_loss_scale * l if task == "traj" else l
            total = l if total is None else total + l
    return total, per_task


# --------------------------------------------------------------------------------------
# optimizer and schedules (vilbert/optimization.py, vilbert/vilbert_init.py)
# --------------------------------------------------------------------------------------
NO_DECAY = ("bias", "LayerNorm.weight", "LayerNorm.bias")   # vilbert_init.py:9 -- substring match


def decays(name: str) -> bool:
    """True when `name` falls in the weight-decay group (vilbert_init.py:14-18).
    Note biOutput.LayerNorm1.weight / LayerNorm2.weight do NOT match the substrings and ARE decayed."""
    return not any(nd in name for nd in NO_DECAY)


@dataclass
class AdamWState:
    step: Dict[str, int] = field(default_factory=dict)
    exp_avg: State = field(default_factory=dict)
    exp_avg_sq: State = field(default_factory=dict)


def adamw_step(params: State, grads: Dict[str, Optional[Tensor]], st: AdamWState, lr: float,
               weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-6) -> None:
    """AdamW.step, vilbert/optimization.py:141-187 (correct_bias=True). In place on `params`.

    Tensors whose grad is None are skipped entirely -- no state, no decay (:143-144).
    """
    b1, b2 = betas
    with torch.no_grad():
        for name, p in params.items():
            g = grads.get(name)
            if g is None:
                continue
            if name not in st.step:
                st.step[name] = 0
                st.exp_avg[name] = torch.zeros_like(p)
                st.exp_avg_sq[name] = torch.zeros_like(p)
            m, v = st.exp_avg[name], st.exp_avg_sq[name]
            st.step[name] += 1
            m.mul_(b1).add_(g, alpha=1.0 - b1)                      # :166
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)               # :167
            denom = v.sqrt().add_(eps)                              # :168
            t = st.step[name]
            step_size = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)   # :170-174
            p.addcdiv_(m, denom, value=-step_size)                  # :176
            wd = weight_decay if decays(name) else 0.0
            if wd > 0.0:
                p.add_(p, alpha=-lr * wd)                           # :186-187 (after the update)


def warmup_linear(step: int, warmup_steps: float, t_total: float) -> float:
    """WarmupLinearSchedule.lr_lambda, vilbert/optimization.py:57-61."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def schedule_totals(loader_len: int, grad_accum: int, num_epochs: int, warmup_proportion: float = 0.2,
                    cooldown_factor: float = 2.0) -> Tuple[float, float]:
    """vilbert/vilbert_init.py:26-30 -> (warmup_steps, adjusted_t_total)."""
    t_total = (loader_len // grad_accum) * num_epochs
    warm = warmup_proportion * t_total
    return warm, warm + cooldown_factor * (t_total - warm)


# --------------------------------------------------------------------------------------
# helpers shared by tests / bench (not part of the reference's behaviour)
# --------------------------------------------------------------------------------------
def trainable(S: State) -> State:
    """Leaf copies with requires_grad for autograd through the functional model. The tied decoder key
    (vilbert.py:901) is aliased to the word-embedding leaf so its gradient accumulates there (H6)."""
    out: State = {}
    for k, v in S.items():
        if k == "cls.predictions.decoder.weight":
            continue
        out[k] = v.detach().clone().requires_grad_(True)
    if "bert.embeddings.word_embeddings.weight" in out and "cls.predictions.bias" in out:
        out["cls.predictions.decoder.weight"] = out["bert.embeddings.word_embeddings.weight"]
    return out


def train_step(S: State, cfg: RefConfig, flags: TaskFlags, batch, st: AdamWState, lr: float,
               weight_decay: float = 0.01, drop: bool = False):
    """One body of train_epoch (utils/utils_init.py:199-239, grad_accum = 1): fwd, losses, bwd, AdamW."""
    W = trainable(S)
    ids, feat, loc, seg, imask, vmask = model_input(batch)
    out = lily_forward(W, cfg, flags, ids, feat, loc, seg, imask, vmask, drop=drop)
    loss, per_task = total_loss(batch, out, flags)
    loss.backward()
    names = [k for k in W if k != "cls.predictions.decoder.weight"]
    grads = {k: W[k].grad for k in names}
    params = {k: S[k] for k in names}
    adamw_step(params, grads, st, lr, weight_decay)
    if "cls.predictions.decoder.weight" in S:
        S["cls.predictions.decoder.weight"] = S["bert.embeddings.word_embeddings.weight"]
    return loss.detach(), {k: v.detach() for k, v in per_task.items()}, grads, out


# --------------------------------------------------------------------------------------
# batch preparation on the host side of the reference (utils/dataset/common.py:213-300), restated with the uniform draws as
# explicit inputs so that the HIP kernels (csrc/batch.hip) can be pinned bit-for-bit
# --------------------------------------------------------------------------------------
def randomize_tokens(tokens: Tensor, mask: Tensor, p_raw: Tensor, random_tokens: Tensor, mask_token_id: int = 103):
    """common.py:213-270 with mask_action_rate == 0.  p_raw = torch.rand_like(tokens.float()); returns (tokens, targets)."""
    tokens = tokens.clone()
    targets = torch.ones_like(tokens) * -1
    p = p_raw * mask.float()
    thresh = 0.85
    targets[p >= thresh] = tokens[p >= thresh]
    tokens[p >= thresh] = mask_token_id                     # 80 %: [MASK]
    thresh = 0.85 + 0.15 * 0.8
    tokens[p >= thresh] = random_tokens[p >= thresh]        # 10 %: random word
    thresh = 0.85 + 0.15 * 0.9
    tokens[p >= thresh] = targets[p >= thresh]              # 10 %: unchanged
    return tokens, targets


def randomize_regions(features: Tensor, probs: Tensor, mask: Tensor, p_raw: Tensor):
    """common.py:272-300.  p_raw = torch.rand_like(mask.float()); returns (features, targets, targets_mask)."""
    features = features.clone()
    targets = torch.ones_like(probs) / probs.shape[-1]
    targets_mask = torch.zeros_like(mask)
    p = p_raw * mask.float()
    thresh = 0.85
    targets[p >= thresh] = probs[p >= thresh]
    targets_mask[p >= thresh] = 1
    thresh = 0.85 + 0.15 * 0.1
    features[p >= thresh] = 0
    return features, targets, targets_mask
