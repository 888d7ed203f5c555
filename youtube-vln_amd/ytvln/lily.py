"""`Lily` task wrapper (reference `lily.py:23-129`): BertModel + pre-training heads + ranking / trajectory-judgement logits.

Same constructor (`Lily(config, dropout_prob=0.1)` with the task flags on `config.args`), same `forward` signature and the
same outputs dict keyed `ranking` / `traj` / `vision` / `language` (only flagged keys present).  Heads whose output the
reference computes and then discards (`lily.py:87-89` always runs the 30522-way and the 1601-way decoders) are skipped
when their flag is off -- the returned dict is identical.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .vilbert import BertConfig as ViLBertConfig
from .vilbert import BertModel as ViLBertModel
from .vilbert import BertPreTrainedModel as PreTrainedModel
from .vilbert import BertPreTrainingHeads as ViLBertPreTrainingHeads
from .vilbert import _drop_state

BERT_CONFIG_FACTORY = {"vilbert": ViLBertConfig}
BERT_MODEL_FACTORY = {"vilbert": ViLBertModel}
CLS_MODEL_FACTORY = {"vilbert": ViLBertPreTrainingHeads}


class Lily(PreTrainedModel):
    def __init__(self, config, dropout_prob=0.1):
        super().__init__(config)
        self.args = config.args
        if self.args.model_name != "vilbert":
            raise NotImplementedError(f"model_name={self.args.model_name!r}: only 'vilbert' exists in the reference factories")
        self.bert = BERT_MODEL_FACTORY[self.args.model_name](config)
        self.cls = CLS_MODEL_FACTORY[self.args.model_name](config, self.bert.embeddings.word_embeddings.weight)
        bi_hidden_size = config.bi_hidden_size
        self.vil_logit = torch.nn.Linear(bi_hidden_size, 1)
        self.judge = torch.nn.Linear(bi_hidden_size, 1)
        self.dropout = torch.nn.Dropout(dropout_prob)
        self.fusion_method = config.fusion_method
        self.apply(self.init_bert_weights)

    def forward(self, instr_tokens, image_features, image_locations, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, co_attention_mask=None, highlight_tokens=None,
                order_atteneded_visual_feature=None, head_rows=None) -> Dict[str, torch.Tensor]:
        """`head_rows` (extension, default off): {"language": idx, "vision": idx} -> the corresponding outputs hold logits for
        those rows only ([len(idx), vocab]); used by ytvln.utils_init.train_step(loss_aware_heads=True)."""
        sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, _ = self.bert(
            input_txt=instr_tokens, input_imgs=image_features, image_loc=image_locations, token_type_ids=token_type_ids,
            attention_mask=attention_mask, image_attention_mask=image_attention_mask, co_attention_mask=co_attention_mask,
            output_all_encoded_layers=False)

        want = tuple(h for h, on in (("t", self.args.masked_language), ("v", self.args.masked_vision)) if on)
        rows = None
        if head_rows:
            rows = {k: head_rows[n] for k, n in (("t", "language"), ("v", "vision")) if n in head_rows}
        linguistic_prediction, vision_prediction, _ = self.cls(sequence_output_t, sequence_output_v, pooled_output_t,
                                                               pooled_output_v, heads=want, rows=rows)

        if self.fusion_method == "sum":
            pooled_output = pooled_output_t + pooled_output_v
        elif self.fusion_method == "mul":
            pooled_output = pooled_output_t * pooled_output_v
        else:
            assert False
        pooled_output = ops.dropout(pooled_output, self.dropout.p, self.training, _drop_state(self, pooled_output))

        outputs: Dict[str, torch.Tensor] = {}
        if self.args.ranking:
            outputs["ranking"] = ops.linear(pooled_output, self.vil_logit.weight, self.vil_logit.bias)
        if self.args.traj_judge:
            outputs["traj"] = ops.linear(pooled_output, self.judge.weight, self.judge.bias)
        if self.args.masked_vision:
            outputs["vision"] = vision_prediction
        if self.args.masked_language:
            outputs["language"] = linguistic_prediction
        return outputs
