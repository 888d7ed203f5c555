"""On-device batch preparation: the masking the reference's datasets apply per item on the host
(`utils/dataset/common.py:213-300`, called from `utils/dataset/all_dataset.py`), done for a whole `[bs, K, ...]` batch that already
sits in HBM.  The loader then only has to ship the un-masked tokens / features / class probabilities once.

    randomize_tokens(tokens, mask)                   -> (tokens, targets)                  common.py:213-270
    randomize_regions(features, probs, mask)         -> (features, targets, targets_mask)  common.py:272-300   (features in place)
    mask_batch(batch)                                -> the 16-tuple with items 1, 4, 5, 6, 8 replaced

Draws come from the library's Philox stream (`ops.DropoutState`: seed + device-side counter, so the masks change every call and
replay inside a hipGraph); passing `p=` / `random_tokens=` reproduces the reference functions bit for bit (tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import call

MASK_TOKEN_ID = 103         # tokenizer.vocab["[MASK]"] of bert-base-uncased (common.py:255)
VOCAB_SIZE = 30522          # len(tokenizer.vocab)


def _dev(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"ytvln.batch: `{name}` must live on the GPU (no CPU path)")
    return t.contiguous()


def randomize_tokens(tokens: Tensor, mask: Tensor, vocab_size: int = VOCAB_SIZE, mask_token_id: int = MASK_TOKEN_ID,
                     p: Optional[Tensor] = None, random_tokens: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    tokens = _dev(tokens, "tokens").long()
    mask = _dev(mask, "mask").long()
    out, targets = torch.empty_like(tokens), torch.empty_like(tokens)
    explicit = p is not None and random_tokens is not None
    st = None if explicit else ops.DropoutState(tokens.device)
    call("ytvln_randomize_tokens", ops._ptr(tokens), ops._ptr(mask), tokens.numel(), int(vocab_size), int(mask_token_id),
         ops._ptr(_dev(p, "p").float()) if explicit else None, ops._ptr(_dev(random_tokens, "random_tokens").long()) if explicit else None,
         None if explicit else ops._ptr(st.tensor), 0 if explicit else st.next_site(), ops._ptr(out), ops._ptr(targets), ops._stream())
    return out, targets


def randomize_regions(features: Tensor, probs: Tensor, mask: Tensor, p: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """`features` ([..., F], fp32, unit inner stride) is modified IN PLACE like the reference does (`features[p >= thresh] = 0`)."""
    if not (features.is_cuda and features.dtype == torch.float32 and features.is_contiguous()):
        raise RuntimeError("ytvln.batch: `features` must be a contiguous fp32 GPU tensor (it is masked in place)")
    probs = _dev(probs, "probs").float()
    mask = _dev(mask, "mask").long()
    F, C = features.shape[-1], probs.shape[-1]
    rows = mask.numel()
    assert features.numel() == rows * F and probs.numel() == rows * C, (features.shape, probs.shape, mask.shape)
    targets = torch.empty_like(probs)
    tmask = torch.empty_like(mask)
    st = None if p is not None else ops.DropoutState(features.device)
    call("ytvln_randomize_regions", ops._ptr(features), F, ops._ptr(probs), ops._ptr(mask), rows, F, C,
         ops._ptr(_dev(p, "p").float()) if p is not None else None, None if p is not None else ops._ptr(st.tensor),
         0 if p is not None else st.next_site(), ops._ptr(targets), ops._ptr(tmask), ops._stream())
    return features, targets, tmask


def expand_options(pool_features: Tensor, pool_boxes: Tensor, pool_probs: Tensor, pool_masks: Tensor, index: Tensor):
    """Option expansion on the device.  The reference materialises K options per item on the host -- the positive path, caption
    negatives that SHARE its features, frame permutations of it and paths with some frames swapped for random photos
    (`utils/dataset/all_dataset.py:185-233`, `common.py:430-522`) -- and ships all K copies (132 MB of features per step at
    BASELINE config 2).  Here the loader ships each distinct frame once:

        pool_features [P, boxes, F]   pool_boxes [P, boxes, 12]   pool_probs [P, boxes, C]   pool_masks [P, boxes]  (one entry per frame)
        index         [bs, K, frames] int64: the pool entry shown at each position of each option, -1 = padding frame

    and the [bs, K, frames*boxes, ...] tensors of the 16-tuple are gathered in HBM (column 11 of the boxes = the position of the
    frame inside the option, `all_dataset.py:314`).  Returns (image_features, image_boxes, image_probs, image_masks)."""
    bs, K, frames = index.shape
    P, boxes = pool_masks.shape
    idx = _dev(index, "index").long().reshape(-1)
    n = idx.numel()
    feats = ops.gather_rows(_dev(pool_features, "pool_features").reshape(P, -1), idx).view(bs, K, frames * boxes, -1)
    bx = ops.gather_rows(_dev(pool_boxes, "pool_boxes").reshape(P, -1), idx).view(bs, K, frames, boxes, -1)
    valid = (idx >= 0).view(bs, K, frames, 1)
    pos = torch.arange(frames, device=idx.device, dtype=bx.dtype).view(1, 1, frames, 1).expand(bs, K, frames, boxes)
    bx[..., 11] = torch.where(valid.expand(bs, K, frames, boxes), pos, torch.zeros_like(pos))
    probs = ops.gather_rows(_dev(pool_probs, "pool_probs").reshape(P, -1), idx).view(bs, K, frames * boxes, -1)
    masks = (_dev(pool_masks, "pool_masks").long()[idx.clamp(min=0)] * (idx >= 0).view(n, 1)).view(bs, K, frames * boxes)
    return feats, bx.view(bs, K, frames * boxes, -1), probs, masks


def mask_batch(batch: Sequence[Tensor], vocab_size: int = VOCAB_SIZE, mask_token_id: int = MASK_TOKEN_ID) -> List[Tensor]:
    """Apply both maskings to an un-masked device batch in the 16-tuple layout of `get_model_input` (utils/utils_init.py:34-53):
    1 image_features, 3 image_masks, 4 image_targets (class probabilities in, MVM targets out), 5 image_targets_mask (out),
    6 instr_tokens, 7 instr_mask, 8 instr_targets (out)."""
    b = list(batch)
    b[6], b[8] = randomize_tokens(b[6], b[7], vocab_size, mask_token_id)
    b[1], b[4], b[5] = randomize_regions(b[1], b[4], b[3])
    return b
