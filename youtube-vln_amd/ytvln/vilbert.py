"""ViLBERT two-stream co-attentional transformer on hand-written CDNA4 kernels.

Drop-in counterpart of the reference `vilbert/vilbert.py`: the same class names, constructor arguments, `forward`
signatures / return structures and state-dict keys (tests/golden/state_dict_schema.json is dumped from the reference), so
`pretrain.py` / `train.py` / `lily.py` can import these classes instead of the reference's.  The modules only *hold*
parameters (nn.Linear / nn.Embedding containers are created in the reference's order so that seeded initialisation draws
the same numbers); all arithmetic goes through `ytvln.ops` -> libytvln.so:

  reference (vilbert/vilbert.py)                         here
  ------------------------------------------------------ ---------------------------------------------------------------
  BertEmbeddings.forward            :240-256             one fused gather+add+LayerNorm+dropout kernel
  BertImageEmbeddings.forward       :1356-1370           MFMA GEMM (2048->Hv) + one fused loc/orient/frame+LN+dropout kernel
  Bert(Image)SelfAttention.forward  :284-311 / :413-440  one packed QKV GEMM + fused flash attention (scores stay in registers)
  Bert(Image)SelfOutput / *Output   :321-325 / :364-368  GEMM + fused dropout+residual+LayerNorm kernel
  Bert(Image)Intermediate           :351-354             GEMM with erf-GELU epilogue (fused into ops.ffn with the next GEMM)
  BertBiAttention.forward           :552-618             per-direction packed K|V GEMMs + the same attention kernel both ways
  heads                             :851-969             GEMM(+GELU) + LayerNorm kernel + decoder GEMM

GPU only: tensors must be on a HIP device; there is no CPU path (use `oracle/` for CPU checks -- tests only).
"""
from __future__ import annotations

import copy
import json
import logging
import math
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from . import ops
from .ops import DropoutState

logger = logging.getLogger(__name__)


@dataclass
class BertConfig:
    """Same fields / defaults / strictness as the reference dataclass (vilbert/vilbert.py:129-195)."""
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
    v_biattention_id: Tuple[int, ...] = (0, 1)
    t_biattention_id: Tuple[int, ...] = (10, 11)
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

    def __post_init__(self):
        assert len(self.v_biattention_id) == len(self.t_biattention_id)
        assert max(self.v_biattention_id) < self.v_num_hidden_layers
        assert max(self.t_biattention_id) < self.num_hidden_layers

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as fid:
            return cls(**json.load(fid))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps({k: v for k, v in self.to_dict().items() if k != "args"}, indent=2, sort_keys=True, default=str) + "\n"

    def __repr__(self):
        return str(self.to_json_string())


# ------------------------------------------------------------------------------------------------------------------
# per-forward runtime context (dropout key material); set by the outermost model forward
# ------------------------------------------------------------------------------------------------------------------
class _Runtime:
    drop: Optional[DropoutState] = None


def _begin_forward(module: nn.Module, device) -> None:
    _Runtime.drop = DropoutState(device) if module.training else None


def _drop_state(module: nn.Module, x: torch.Tensor) -> Optional[DropoutState]:
    if not module.training:
        return None
    if _Runtime.drop is None:
        # (a module used on its own, e.g. a bare BertEncoder: no _begin_forward ran.)  Inside a two-stream region the clone of the seed and the
        # counter increment must be ordered before BOTH sides' first read: enqueue them on the main stream and make that the new fork point,
        # whichever side asks first (created on the side stream, the main stream could read the state before it was written).
        with ops.TwoStream.shared_write():
            _Runtime.drop = DropoutState(x.device)
        _share_drop_state()
    return _Runtime.drop


def _share_drop_state() -> None:
    """Inside a two-stream region the (seed, counter) tensor of this forward is read by kernels of both HIP streams."""
    ts = ops.TwoStream
    if ts.active and _Runtime.drop is not None:
        t = _Runtime.drop.tensor
        t.record_stream(ts.side_stream(t.device))
        if ts.main is not None:
            t.record_stream(ts.main)


def _p(module: nn.Module, p: float) -> float:
    return float(p) if module.training else 0.0


def gelu(x: torch.Tensor) -> torch.Tensor:
    """vilbert.py:113-119: the exact (erf) GELU.  Inside the model the activation is fused into the GEMM epilogue (ops.linear(..., "gelu"));
    this stand-alone form exists for callers that import it."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def swish(x: torch.Tensor) -> torch.Tensor:
    """vilbert.py:122-123."""
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": gelu, "relu": torch.nn.functional.relu, "swish": swish}        # vilbert.py:126


def load_tf_weights_in_bert(model, tf_checkpoint_path):
    """vilbert.py:58-110 reads a TensorFlow checkpoint through the `tensorflow` package.  Not part of the hot path and not available
    here: convert the checkpoint to a PyTorch state dict once (the reference's own loader does that) and use `from_pretrained`."""
    raise NotImplementedError("TensorFlow checkpoints are not read by this build; convert to a PyTorch state dict and use from_pretrained()")


def _act_name(act) -> str:
    if not isinstance(act, str) or act not in ("gelu", "relu"):
        raise NotImplementedError(f"activation {act!r}: the HIP path implements 'gelu' (erf form) and 'relu'")
    return act


def _mask2d(mask: torch.Tensor, n: int, t: int) -> torch.Tensor:
    """Additive attention mask as [N, T] fp32.  The reference passes it extended to [N,1,1,T] (vilbert.py:1268-1287)."""
    if mask.numel() != n * t:
        raise NotImplementedError(f"attention mask of shape {tuple(mask.shape)}: only per-key masks ([N,1,1,T]) are supported")
    m = mask.reshape(n, t)
    if m.dtype != torch.float32:
        m = m.float()
    return m.contiguous()


class BertLayerNorm(nn.Module):
    """TF-style LayerNorm, eps inside the sqrt (vilbert.py:204-217) -- fused kernel."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.add_layer_norm(x, None, self.weight, self.bias, self.variance_epsilon)


def _add_ln(ln: BertLayerNorm, x, res, module, p_pre=0.0, p_post=0.0):
    return ops.add_layer_norm(x, res, ln.weight, ln.bias, ln.variance_epsilon, _p(module, p_pre), _p(module, p_post),
                              _drop_state(module, x))


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None):
        return ops.text_embed(input_ids, token_type_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                              self.token_type_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias,
                              self.LayerNorm.variance_epsilon, _p(self, self.dropout.p), _drop_state(self, self.LayerNorm.weight))


class _PackedQKV:
    """Helper: one [3H, Hin] projection instead of three (the reference runs query/key/value as separate Linears)."""

    @staticmethod
    def project(x, q: nn.Linear, k: nn.Linear, v: nn.Linear):
        return ops.linear(x, ops.pack_rows(q.weight, k.weight, v.weight), ops.pack_rows(q.bias, k.bias, v.bias))

    @staticmethod
    def project_res(x, q: nn.Linear, k: nn.Linear, v: nn.Linear):
        """(q|k|v projection, x for the residual around the attention block): see ops.LinearFn(passthrough)."""
        return ops.linear_res(x, ops.pack_rows(q.weight, k.weight, v.weight), ops.pack_rows(q.bias, k.bias, v.bias))


class _SelfAttentionBase(nn.Module):
    """Shared forward of BertSelfAttention / BertImageSelfAttention."""

    want_probs = False

    def _build(self, hidden, heads, p_attn):
        if hidden % heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (hidden, heads))
        self.num_attention_heads = heads
        self.attention_head_size = int(hidden / heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(hidden, self.all_head_size)
        self.key = nn.Linear(hidden, self.all_head_size)
        self.value = nn.Linear(hidden, self.all_head_size)
        self.dropout = nn.Dropout(p_attn)

    def forward(self, hidden_states, attention_mask):
        out, probs, _ = self.forward_res(hidden_states, attention_mask)
        return out, probs

    def forward_res(self, hidden_states, attention_mask):
        """(context, probs, hidden_states-for-the-residual): the third value is what the caller's `LayerNorm(dense(context) + input)`
        must use as `input` so that the two gradients of `hidden_states` meet inside the projection's input-gradient GEMM."""
        n, t, _ = hidden_states.shape
        mask = _mask2d(attention_mask, n, t)
        qkv, residual = _PackedQKV.project_res(hidden_states, self.query, self.key, self.value)
        qkv = qkv.view(n * t, 3 * self.all_head_size)
        p = _p(self, self.dropout.p)
        st = _drop_state(self, qkv) if p > 0 else None
        out, lse = ops.SelfAttentionFn.apply(qkv, mask, n, t, self.num_attention_heads, p, st.tensor if st else None,
                                             st.next_site() if st else 0)
        probs = None
        if self.want_probs:
            with torch.no_grad():
                h = self.all_head_size
                probs = ops.attn_probs(qkv, 0, 3 * h, qkv, h, 3 * h, mask, lse, n, self.num_attention_heads, t, t,
                                       self.attention_head_size, 1.0 / math.sqrt(self.attention_head_size))
        return out.view(n, t, self.all_head_size), probs, residual


class BertSelfAttention(_SelfAttentionBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.hidden_size, config.num_attention_heads, config.attention_probs_dropout_prob)


class _SelfOutputBase(nn.Module):
    def _build(self, hidden, p):
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = BertLayerNorm(hidden, eps=1e-12)
        self.dropout = nn.Dropout(p)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        return _add_ln(self.LayerNorm, h, input_tensor, self, p_pre=self.dropout.p)


class BertSelfOutput(_SelfOutputBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.hidden_size, config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output, attention_probs, residual = self.self.forward_res(input_tensor, attention_mask)
        return self.output(self_output, residual), attention_probs


class _IntermediateBase(nn.Module):
    def _build(self, hidden, inter, act):
        self.dense = nn.Linear(hidden, inter)
        self.intermediate_act_fn = act

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, _act_name(self.intermediate_act_fn))


class BertIntermediate(_IntermediateBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.hidden_size, config.intermediate_size, config.hidden_act)


class _OutputBase(nn.Module):
    def _build(self, inter, hidden, p):
        self.dense = nn.Linear(inter, hidden)
        self.LayerNorm = BertLayerNorm(hidden, eps=1e-12)
        self.dropout = nn.Dropout(p)

    def forward(self, hidden_states, input_tensor):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        return _add_ln(self.LayerNorm, h, input_tensor, self, p_pre=self.dropout.p)


class BertOutput(_OutputBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.intermediate_size, config.hidden_size, config.hidden_dropout_prob)


def _ffn_block(inter: _IntermediateBase, out: _OutputBase, x):
    """intermediate -> output of a layer, with the two GEMMs and the GELU fused in one autograd node when possible."""
    if _act_name(inter.intermediate_act_fn) == "gelu":
        h, residual = ops.ffn_res(x, inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias)
        return _add_ln(out.LayerNorm, h, residual, out, p_pre=out.dropout.p)
    return out(inter(x), x)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask):
        attention_output, attention_probs = self.attention(hidden_states, attention_mask)
        return _ffn_block(self.intermediate, self.output, attention_output), attention_probs


class BertImageSelfAttention(_SelfAttentionBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.v_hidden_size, config.v_num_attention_heads, config.v_attention_probs_dropout_prob)


class BertImageSelfOutput(_SelfOutputBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.v_hidden_size, config.v_hidden_dropout_prob)


class BertImageAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertImageSelfAttention(config)
        self.output = BertImageSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output, attention_probs, residual = self.self.forward_res(input_tensor, attention_mask)
        return self.output(self_output, residual), attention_probs


class BertImageIntermediate(_IntermediateBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.v_hidden_size, config.v_intermediate_size, config.v_hidden_act)


class BertImageOutput(_OutputBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.v_intermediate_size, config.v_hidden_size, config.v_hidden_dropout_prob)


class BertImageLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertImageAttention(config)
        self.intermediate = BertImageIntermediate(config)
        self.output = BertImageOutput(config)

    def forward(self, hidden_states, attention_mask):
        attention_output, attention_probs = self.attention(hidden_states, attention_mask)
        return _ffn_block(self.intermediate, self.output, attention_output), attention_probs


class BertBiAttention(nn.Module):
    """Cross-stream co-attention (vilbert.py:512-618).  stream 1 = vision, stream 2 = text."""

    want_probs = False

    def __init__(self, config):
        super().__init__()
        if config.bi_hidden_size % config.bi_num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.bi_hidden_size, config.bi_num_attention_heads))
        self.num_attention_heads = config.bi_num_attention_heads
        self.attention_head_size = int(config.bi_hidden_size / config.bi_num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.key1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.value1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.dropout1 = nn.Dropout(config.v_attention_probs_dropout_prob)
        self.query2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.key2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.value2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout2 = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        """The reference's 3-tuple (vilbert.py:618)."""
        return self.forward_res(input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask, use_co_attention_mask)[:3]

    def forward_res(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                    use_co_attention_mask=False):
        """forward() plus the two stream inputs as they leave the projections' autograd nodes (the residual values BertBiOutput adds):
        returned, not parked on the module -- no graph stays pinned between calls and a direct forward() cannot leave a stale pair behind
        (ADVICE r2).  BertConnectionLayer calls this form."""
        if use_co_attention_mask:
            raise NotImplementedError("use_co_attention_mask=True is dead code in the reference (vilbert.py:736) and unsupported here")
        n, r, _ = input_tensor1.shape
        t = input_tensor2.shape[1]
        hb = self.all_head_size
        m1, m2 = _mask2d(attention_mask1, n, r), _mask2d(attention_mask2, n, t)
        # each stream's input feeds two projections and the residual of BertBiOutput: the residual value travels through both
        # projections' autograd nodes (ops.LinearFn passthrough) so the three gradients meet inside the input-gradient GEMMs
        q1, res1 = ops.linear_res(input_tensor1, self.query1.weight, self.query1.bias)
        kv1, res1 = ops.linear_res(res1, ops.pack_rows(self.key1.weight, self.value1.weight), ops.pack_rows(self.key1.bias, self.value1.bias))
        with ops.TwoStream.side(input_tensor2):         # the text side's projections: second HIP stream when two-stream mode is on
            q2, res2 = ops.linear_res(input_tensor2, self.query2.weight, self.query2.bias)
            kv2, res2 = ops.linear_res(res2, ops.pack_rows(self.key2.weight, self.value2.weight), ops.pack_rows(self.key2.bias, self.value2.bias))
        ops.TwoStream.join(q2, kv2)
        q1, kv1, q2, kv2 = q1.view(n * r, hb), kv1.view(n * r, 2 * hb), q2.view(n * t, hb), kv2.view(n * t, 2 * hb)
        p1, p2 = _p(self, self.dropout1.p), _p(self, self.dropout2.p)
        st = _drop_state(self, q1) if (p1 > 0 or p2 > 0) else None
        s1, s2 = (st.next_site(), st.next_site()) if st else (0, 0)
        ctx1, ctx2, lse1, lse2 = ops.CoAttentionFn.apply(q1, kv1, q2, kv2, m1, m2, n, r, t, self.num_attention_heads, p1, p2,
                                                         st.tensor if st else None, s1, s2)
        ops.TwoStream.mark()                            # both context tensors exist: the text side may go on from here
        probs = (None, None)
        if self.want_probs:
            with torch.no_grad():
                sc = 1.0 / math.sqrt(self.attention_head_size)
                pr1 = ops.attn_probs(q2, 0, hb, kv1, 0, 2 * hb, m1, lse1, n, self.num_attention_heads, t, r,
                                     self.attention_head_size, sc)
                pr2 = ops.attn_probs(q1, 0, hb, kv2, 0, 2 * hb, m2, lse2, n, self.num_attention_heads, r, t,
                                     self.attention_head_size, sc)
                probs = (pr1, pr2)
        return ctx1.view(n, t, hb), ctx2.view(n, r, hb), probs, res1, res2


class BertBiOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense1 = nn.Linear(config.bi_hidden_size, config.v_hidden_size)
        self.LayerNorm1 = BertLayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout1 = nn.Dropout(config.v_hidden_dropout_prob)
        self.q_dense1 = nn.Linear(config.bi_hidden_size, config.v_hidden_size)   # declared, never used (vilbert.py:628)
        self.q_dropout1 = nn.Dropout(config.v_hidden_dropout_prob)
        self.dense2 = nn.Linear(config.bi_hidden_size, config.hidden_size)
        self.LayerNorm2 = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout2 = nn.Dropout(config.hidden_dropout_prob)
        self.q_dense2 = nn.Linear(config.bi_hidden_size, config.hidden_size)     # declared, never used (vilbert.py:635)
        self.q_dropout2 = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states1, input_tensor1, hidden_states2, input_tensor2):
        c1 = ops.linear(hidden_states1, self.dense1.weight, self.dense1.bias)
        with ops.TwoStream.side(hidden_states2, input_tensor2):
            c2 = ops.linear(hidden_states2, self.dense2.weight, self.dense2.bias)
        h1 = _add_ln(self.LayerNorm1, c1, input_tensor1, self, p_pre=self.dropout1.p)
        with ops.TwoStream.side():
            h2 = _add_ln(self.LayerNorm2, c2, input_tensor2, self, p_pre=self.dropout2.p)
        return h1, h2


class BertConnectionLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.biattention = BertBiAttention(config)
        self.biOutput = BertBiOutput(config)
        self.v_intermediate = BertImageIntermediate(config)
        self.v_output = BertImageOutput(config)
        self.t_intermediate = BertIntermediate(config)
        self.t_output = BertOutput(config)

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        bi_output1, bi_output2, co_attention_probs, res1, res2 = self.biattention.forward_res(
            input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask, use_co_attention_mask)
        # bi_output2 (image queries over text) feeds the vision stream, bi_output1 the text stream (vilbert.py:671)
        attention_output1, attention_output2 = self.biOutput(bi_output2, res1, bi_output1, res2)
        layer_output1 = _ffn_block(self.v_intermediate, self.v_output, attention_output1)
        with ops.TwoStream.side():
            layer_output2 = _ffn_block(self.t_intermediate, self.t_output, attention_output2)
        return layer_output1, layer_output2, co_attention_probs


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.FAST_MODE = config.fast_mode
        self.with_coattention = config.with_coattention
        self.v_biattention_id = config.v_biattention_id
        self.t_biattention_id = config.t_biattention_id
        self.in_batch_pairs = config.in_batch_pairs
        self.fixed_t_layer = config.fixed_t_layer
        self.fixed_v_layer = config.fixed_v_layer
        layer = BertLayer(config)
        v_layer = BertImageLayer(config)
        connect_layer = BertConnectionLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(config.num_hidden_layers)])
        self.v_layer = nn.ModuleList([copy.deepcopy(v_layer) for _ in range(config.v_num_hidden_layers)])
        self.c_layer = nn.ModuleList([copy.deepcopy(connect_layer) for _ in range(len(config.v_biattention_id))])
        # Autograd cut points for a backward pass that runs in phases (ytvln.distributed.GraphedTrainStep, mode "phased": the gradients
        # of the layers above a cut are exchanged between the ranks while the layers below it are still running their backward).
        # Names: "t<i>" / "v<i>" = after text / image layer i, "c<i>" = after co-attention layer i.  At a cut the hidden states are
        # replaced by detached leaves; `_cuts` keeps (outputs below the cut, leaves above it) in forward order.  Values are unchanged.
        self.cut_after = frozenset()
        self._cuts = []

    def _cut(self, name, *hidden):
        if name not in self.cut_after or not torch.is_grad_enabled() or not any(h.requires_grad for h in hidden):
            return hidden if len(hidden) > 1 else hidden[0]
        below = [h for h in hidden if h.requires_grad]
        above = [h.detach().requires_grad_() for h in below]
        self._cuts.append((name, below, above))
        it = iter(above)
        out = tuple(next(it) if h.requires_grad else h for h in hidden)
        return out if len(out) > 1 else out[0]

    def _set_probs(self, flag: bool):
        for m in self.modules():
            if isinstance(m, (_SelfAttentionBase, BertBiAttention)):
                m.want_probs = flag

    def forward(self, txt_embedding, image_embedding, txt_attention_mask, image_attention_mask, co_attention_mask=None,
                output_all_encoded_layers=True, output_all_attention_masks=False):
        """Interleaving schedule of vilbert.py:737-811: for each (v_id, t_id) pair run the pending image layers, the
        pending text layers, then co-attention layer `count`; finally the remaining layers of both streams."""
        self._set_probs(bool(output_all_attention_masks))
        ts = ops.TwoStream
        if self.in_batch_pairs or self.FAST_MODE:
            ts.end(txt_embedding)                        # these modes rebuild the text rows from the image batch: one stream
        else:
            ts.begin(txt_embedding.device)               # (no-op when BertModel.forward opened the region or the mode is off)
        try:
            return self._forward(txt_embedding, image_embedding, txt_attention_mask, image_attention_mask, co_attention_mask,
                                 output_all_encoded_layers, output_all_attention_masks)
        finally:
            ts.active = False

    def _forward(self, txt_embedding, image_embedding, txt_attention_mask, image_attention_mask, co_attention_mask,
                 output_all_encoded_layers, output_all_attention_masks):
        ts = ops.TwoStream
        v_start = t_start = 0
        all_t, all_v, att_t, att_v, att_c = [], [], [], [], []
        self._cuts = []
        cuts = bool(self.cut_after) and not (self.in_batch_pairs or self.FAST_MODE)

        def run(layers, lo, hi, x, mask, sink, frozen_to, tag):
            for idx in range(lo, hi):
                if idx < frozen_to:
                    with torch.no_grad():
                        x, pr = layers[idx](x, mask)
                else:
                    x, pr = layers[idx](x, mask)
                if cuts:
                    x = self._cut(f"{tag}{idx}", x)
                if output_all_attention_masks:
                    sink.append(pr)
            return x

        for count, (v_end, t_end) in enumerate(zip(self.v_biattention_id, self.t_biattention_id)):
            assert self.fixed_t_layer <= t_end and self.fixed_v_layer <= v_end
            image_embedding = run(self.v_layer, v_start, v_end, image_embedding, image_attention_mask, att_v, self.fixed_v_layer, "v")
            with ts.side(txt_embedding, txt_attention_mask):
                txt_embedding = run(self.layer, t_start, t_end, txt_embedding, txt_attention_mask, att_t, self.fixed_t_layer, "t")
            if count == 0 and self.in_batch_pairs:
                # every text of the batch against every image of the batch: B -> B^2 rows (vilbert.py:771-778); row (i, j) = text i, image j
                b, r_, hv = image_embedding.shape
                t_, ht = txt_embedding.shape[1], txt_embedding.shape[2]
                image_embedding = image_embedding.unsqueeze(0).expand(b, b, r_, hv).contiguous().view(b * b, r_, hv)
                image_attention_mask = image_attention_mask.reshape(b, -1).unsqueeze(0).expand(b, b, r_).contiguous().view(b * b, 1, 1, r_)
                txt_embedding = txt_embedding.unsqueeze(1).expand(b, b, t_, ht).contiguous().view(b * b, t_, ht)
                txt_attention_mask = txt_attention_mask.reshape(b, -1).unsqueeze(1).expand(b, b, t_).contiguous().view(b * b, 1, 1, t_)
            if count == 0 and self.FAST_MODE:
                # one text against many images (vilbert.py:780-782): the text rows are broadcast to the image batch
                nb = image_embedding.size(0)
                txt_embedding = txt_embedding.expand(nb, txt_embedding.size(1), txt_embedding.size(2)).contiguous()
                txt_attention_mask = txt_attention_mask.reshape(txt_attention_mask.shape[0], -1).expand(nb, txt_embedding.size(1)).contiguous().view(
                    nb, 1, 1, txt_embedding.size(1))
            if self.with_coattention:
                image_embedding, txt_embedding, co_probs = self.c_layer[count](
                    image_embedding, image_attention_mask, txt_embedding, txt_attention_mask, co_attention_mask, False)
                if output_all_attention_masks:
                    att_c.append(co_probs)
                if cuts:
                    image_embedding, txt_embedding = self._cut(f"c{count}", image_embedding, txt_embedding)
            v_start, t_start = v_end, t_end
            if output_all_encoded_layers:
                all_t.append(txt_embedding)
                all_v.append(image_embedding)
        image_embedding = run(self.v_layer, v_start, len(self.v_layer), image_embedding, image_attention_mask, att_v, 0, "v")
        with ts.side(txt_embedding, txt_attention_mask):
            txt_embedding = run(self.layer, t_start, len(self.layer), txt_embedding, txt_attention_mask, att_t, 0, "t")
        ts.end(txt_embedding, *all_t, *(p for p in att_t if p is not None))
        if not output_all_encoded_layers:
            all_t.append(txt_embedding)
            all_v.append(image_embedding)
        return all_t, all_v, (att_t, att_v, att_c)


class _PoolerBase(nn.Module):
    def _build(self, hidden, bi_hidden):
        self.dense = nn.Linear(hidden, bi_hidden)
        self.activation = nn.ReLU()

    def forward(self, hidden_states):
        # first token of every row, read in place through the GEMM's leading dimension (no gather copy)
        return ops.linear(hidden_states[:, 0], self.dense.weight, self.dense.bias, "relu", out_fp32=True)      # (fp32 out: only matters on the bf16-resident path)


class BertTextPooler(_PoolerBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.hidden_size, config.bi_hidden_size)


class BertImagePooler(_PoolerBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.v_hidden_size, config.bi_hidden_size)


class _HeadTransformBase(nn.Module):
    def _build(self, hidden, act):
        self.dense = nn.Linear(hidden, hidden)
        self.transform_act_fn = act
        self.LayerNorm = BertLayerNorm(hidden, eps=1e-12)

    def forward(self, hidden_states):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, _act_name(self.transform_act_fn))
        return self.LayerNorm(h)


class BertPredictionHeadTransform(_HeadTransformBase):
    def __init__(self, config):
        super().__init__()
        self._build(config.hidden_size, config.hidden_act)


class BertImgPredictionHeadTransform(_HeadTransformBase):
    def __init__(self, config):
        super().__init__()
        # the reference tests config.hidden_act and then applies ACT2FN[config.hidden_act] (vilbert.py:874-879)
        self._build(config.v_hidden_size, config.hidden_act if isinstance(config.hidden_act, str) else config.v_hidden_act)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(bert_model_embedding_weights.size(1), bert_model_embedding_weights.size(0), bias=False)
        self.decoder.weight = bert_model_embedding_weights          # tied to the word embedding (vilbert.py:901)
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))

    def forward(self, hidden_states):
        return ops.linear(self.transform(hidden_states), self.decoder.weight, self.bias)       # bf16 logits on the bf16-resident path (the loss kernels read them)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertOnlyNSPHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.seq_relationship = nn.Linear(config.hidden_size, 2)

    def forward(self, pooled_output):
        return ops.linear(pooled_output, self.seq_relationship.weight, self.seq_relationship.bias)


class BertImagePredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertImgPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.v_hidden_size, config.v_target_size)

    def forward(self, hidden_states):
        return ops.linear(self.transform(hidden_states), self.decoder.weight, self.decoder.bias)


class BertPreTrainingHeads(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)
        self.bi_seq_relationship = nn.Linear(config.bi_hidden_size, 2)
        self.imagePredictions = BertImagePredictionHead(config)
        self.fusion_method = config.fusion_method
        self.dropout = nn.Dropout(0.1)

    def forward(self, sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, heads=("t", "v", "rel"), rows=None):
        """Returns (prediction_scores_t, prediction_scores_v, seq_relationship_score) like vilbert.py:939-954.
        `heads` lets a caller that discards a head (Lily, lily.py:87) skip its GEMMs; skipped entries are None.
        `rows` = {"t": idx, "v": idx} (optional) decodes only the listed rows of the flattened [N*T, H] / [N*R, Hv] sequences
        (loss-aware heads: the scores come back as [len(idx), vocab] instead of [N, T, vocab])."""
        if rows is not None:
            if "t" in rows:
                sequence_output_t = ops.gather_rows(sequence_output_t.reshape(-1, sequence_output_t.shape[-1]), rows["t"])
            if "v" in rows:
                sequence_output_v = ops.gather_rows(sequence_output_v.reshape(-1, sequence_output_v.shape[-1]), rows["v"])
        if self.fusion_method == "sum":
            pooled = pooled_output_t + pooled_output_v
        elif self.fusion_method == "mul":
            pooled = pooled_output_t * pooled_output_v
        else:
            assert False
        scores_t = self.predictions(sequence_output_t) if "t" in heads else None
        rel = None
        if "rel" in heads:
            pooled = ops.dropout(pooled, self.dropout.p, self.training, _drop_state(self, pooled))
            rel = ops.linear(pooled, self.bi_seq_relationship.weight, self.bi_seq_relationship.bias)
        scores_v = self.imagePredictions(sequence_output_v) if "v" in heads else None
        return scores_t, scores_v, rel


class BertPreTrainedModel(nn.Module):
    """Weight initialisation and checkpoint loading (vilbert.py:972-1179)."""

    def __init__(self, config, default_gpu=True, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def init_bert_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, config, default_gpu=True, state_dict=None, cache_dir=None,
                        from_tf=False, *inputs, **kwargs):
        """Instantiate and load a local PyTorch checkpoint (`.bin`, or a directory holding `pytorch_model.bin`).
        Same key handling as the reference: `model_state_dict` unwrapping (:1104), gamma/beta renames (:1119-1129),
        optional `bert.` prefix (:1155-1159), non-strict with missing / unexpected keys logged (:1161-1172).
        Network download (cached_path) and TensorFlow checkpoints are out of scope."""
        if from_tf:
            raise NotImplementedError("TensorFlow checkpoints are out of scope for the HIP path")
        model = cls(config, *inputs, **kwargs)
        if state_dict is None:
            path = str(pretrained_model_name_or_path)
            if os.path.isdir(path):
                path = os.path.join(path, "pytorch_model.bin")
            if not os.path.exists(path):
                raise RuntimeError(f"checkpoint {path} not found (remote archives are not supported)")
            if default_gpu:
                logger.info("loading archive file {}".format(path))
            state_dict = torch.load(path, map_location="cpu")
            if "model_state_dict" in state_dict:
                state_dict = state_dict["model_state_dict"]
            if "state_dict" in dir(state_dict):
                state_dict = state_dict.state_dict()
        state_dict = {k.replace("gamma", "weight").replace("beta", "bias"): v for k, v in state_dict.items()}
        if not hasattr(model, "bert") and any(s.startswith("bert.") for s in state_dict):
            state_dict = {k[len("bert."):]: v for k, v in state_dict.items() if k.startswith("bert.")}
        result = model.load_state_dict(state_dict, strict=False)
        if result.missing_keys and default_gpu:
            logger.info("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, result.missing_keys))
        if result.unexpected_keys and default_gpu:
            logger.info("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, result.unexpected_keys))
        return model


class BertImageEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.image_embeddings = nn.Linear(config.v_feature_size, config.v_hidden_size)
        self.image_location_embeddings = nn.Linear(5, config.v_hidden_size)
        self.image_orientation_embeddings = nn.Linear(4, config.v_hidden_size)
        self.image_next_orientation_embeddings = nn.Linear(2, config.v_hidden_size)
        self.image_sequence_embeddings = nn.Embedding(32, config.v_hidden_size)
        self.LayerNorm = BertLayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, input_loc):
        if input_ids.dtype != torch.float32:
            input_ids = input_ids.float()
        if input_loc.dtype != torch.float32:
            input_loc = input_loc.float()
        img = ops.linear(input_ids, self.image_embeddings.weight, self.image_embeddings.bias)
        return ops.image_embed(img, input_loc, self.image_location_embeddings.weight, self.image_location_embeddings.bias,
                               self.image_orientation_embeddings.weight, self.image_orientation_embeddings.bias,
                               self.image_next_orientation_embeddings.weight, self.image_next_orientation_embeddings.bias,
                               self.image_sequence_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias,
                               self.LayerNorm.variance_epsilon, _p(self, self.dropout.p), _drop_state(self, img))


class BertModel(BertPreTrainedModel):
    """Two-stream encoder; signature and returns of vilbert.py:1242-1337."""

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.v_embeddings = BertImageEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.t_pooler = BertTextPooler(config)
        self.v_pooler = BertImagePooler(config)
        self.apply(self.init_bert_weights)

    def forward(self, input_txt, input_imgs, image_loc, token_type_ids=None, attention_mask=None, image_attention_mask=None,
                co_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        if not input_imgs.is_cuda:
            raise RuntimeError("ytvln BertModel runs on a HIP device only (inputs are on %s); there is no CPU fallback" % input_imgs.device)
        _begin_forward(self, input_imgs.device)
        n, t = input_txt.shape
        r = input_imgs.size(1)
        # additive masks: 0 where attended, -10000 where padded (vilbert.py:1268-1287); kept as [N,1,1,T] like the reference
        if attention_mask is None:
            ext_t = torch.zeros((n, 1, 1, t), dtype=torch.float32, device=input_txt.device)
        else:
            ext_t = ((1.0 - attention_mask.to(torch.float32)) * -10000.0).view(n, 1, 1, t)
        nv = input_imgs.size(0)          # (differs from the text batch only in fast_mode: one text against many images)
        if image_attention_mask is None:
            ext_v = torch.zeros((nv, 1, 1, r), dtype=torch.float32, device=input_txt.device)
        else:
            ext_v = ((1.0 - image_attention_mask.to(torch.float32)) * -10000.0).view(nv, 1, 1, r)
        # co_attention_mask only feeds the dead use_co_attention_mask branch (vilbert.py:736): not materialised.

        ts = ops.TwoStream
        try:
            if not (self.encoder.in_batch_pairs or self.encoder.FAST_MODE):
                ts.begin(input_imgs.device)              # two-stream mode: the text side starts at its embeddings
                _share_drop_state()
            with ts.side(input_txt, token_type_ids, ext_t):
                embedding_output = self.embeddings(input_txt, token_type_ids)
            v_embedding_output = self.v_embeddings(input_imgs, image_loc)
            encoded_layers_t, encoded_layers_v, all_attention_mask = self.encoder(
                embedding_output, v_embedding_output, ext_t, ext_v, None,
                output_all_encoded_layers=output_all_encoded_layers, output_all_attention_masks=output_all_attention_masks)
        finally:
            ts.active = False
        sequence_output_t, sequence_output_v = encoded_layers_t[-1], encoded_layers_v[-1]
        pooled_output_t = self.t_pooler(sequence_output_t)
        pooled_output_v = self.v_pooler(sequence_output_v)
        if not output_all_encoded_layers:
            encoded_layers_t, encoded_layers_v = encoded_layers_t[-1], encoded_layers_v[-1]
        return encoded_layers_t, encoded_layers_v, pooled_output_t, pooled_output_v, all_attention_mask


class BertForMultiModalPreTraining(BertPreTrainedModel):
    """vilbert.py:1373-1455.  Losses run on the fused loss kernels (no host sync)."""

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertModel(config)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        self.apply(self.init_bert_weights)
        self.predict_feature = config.predict_feature
        print("model's option for predict_feature is ", config.predict_feature)

    def forward(self, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None, image_attention_mask=None,
                masked_lm_labels=None, image_label=None, image_target=None, next_sentence_label=None,
                output_all_attention_masks=False):
        sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, all_attention_mask = self.bert(
            input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask,
            output_all_encoded_layers=False, output_all_attention_masks=output_all_attention_masks)
        prediction_scores_t, prediction_scores_v, seq_relationship_score = self.cls(
            sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v)
        if masked_lm_labels is not None and next_sentence_label is not None and image_target is not None:
            n, r, c = prediction_scores_v.shape
            pv = prediction_scores_v[:, 1:].reshape(n * (r - 1), c)                      # region 0 dropped (:1429)
            label = (image_label == 1).reshape(-1)
            if self.predict_feature:
                # MSE feature regression (:1391, 1430-1434): sum((pred - target)^2 over masked regions) / max(#masked * C, 1).  Off in every
                # target config: a handful of elementwise torch kernels on the device, no fused kernel of its own.
                lm = label.to(pv.dtype).unsqueeze(1)
                diff = pv - image_target.reshape(n * (r - 1), c).to(pv.dtype)
                masked_img_loss = (diff * diff * lm).sum() / torch.clamp(lm.sum() * c, min=1.0)
            else:
                # sum(KL * mask) / max(sum(mask), 0): the reference divides by zero when nothing is masked (:1440-1442);
                # the fused kernel clamps the denominator at 1 in that degenerate case.
                masked_img_loss = ops.kl_masked(pv, image_target.reshape(n * (r - 1), c).float(), label)
            masked_lm_loss = ops.cross_entropy(prediction_scores_t.view(-1, self.config.vocab_size), masked_lm_labels.view(-1), -1)
            next_sentence_loss = ops.cross_entropy(seq_relationship_score.view(-1, 2), next_sentence_label.view(-1), -1)
            return masked_lm_loss.unsqueeze(0), masked_img_loss.unsqueeze(0), next_sentence_loss.unsqueeze(0)
        return prediction_scores_t, prediction_scores_v, seq_relationship_score, all_attention_mask
