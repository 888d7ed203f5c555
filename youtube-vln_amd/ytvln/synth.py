"""Deterministic synthetic inputs and weights (numpy only, no torch RNG).

The reference's data layer (LMDB features + captions, `utils/dataset/all_dataset.py`) is out of scope; the hot path
is fed tensors with the same 16-tuple layout that `get_model_input` (`utils/utils_init.py:34-77`) unpacks:

    0 ranking target  i64 [bs]            8  instr_targets  i64 [bs,K,T]  (-1 = ignore)
    1 image_features  f32 [bs,K,R,F]      9  highlights     f32 [bs,K,0]
    2 image_boxes     f32 [bs,K,R,12]     10 segment_ids    i64 [bs,K,T]
    3 image_masks     i64 [bs,K,R]        11 co_attention   i64 [bs,2,R,T] (dead input)
    4 image_targets   f32 [bs,K,R,C]      12 ids            i64 [bs]
    5 image_tgt_mask  i64 [bs,K,R]        13 opt_mask       bool [bs,K]
    6 instr_tokens    i64 [bs,K,T]        14 ordering target i64 [bs]
    7 instr_mask      bool [bs,K,T]       15 flag           i64 [bs]

The same recipe (same seed -> same bytes) is used by `bench.py`, by the GPU parity tests and by
`oracle/gen_golden.py`, so the GPU box can rebuild the exact inputs the committed goldens were made from.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def make_weights(shapes: Dict[str, Sequence[int]], seed: int = 0) -> Dict[str, np.ndarray]:
    """Numpy-seeded weights for a state dict: per sorted key, N(0,1)*0.02; LayerNorm gains 1+0.1n; biases 0.02n.

    The tied decoder key is an alias of the word embedding (vilbert/vilbert.py:901) and receives the same array.
    """
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for k in sorted(shapes):
        if k == "cls.predictions.decoder.weight":
            continue
        n = rs.standard_normal(tuple(shapes[k])).astype(np.float32)
        if "LayerNorm" in k and k.endswith("weight"):
            out[k] = (1.0 + 0.1 * n).astype(np.float32)
        else:
            out[k] = (0.02 * n).astype(np.float32)
    if "cls.predictions.decoder.weight" in shapes:
        out["cls.predictions.decoder.weight"] = out["bert.embeddings.word_embeddings.weight"]
    return out


def make_batch(bs: int, K: int, T: int, frames: int, boxes: int, F: int = 2048, C: int = 1601, vocab: int = 30522,
               seed: int = 1234, finetune_heading: bool = False, opt_holes: int = 0, ignore_rank_frac: float = 0.05,
               min_len: int = 0) -> List[np.ndarray]:
    """Build the 16-tuple (numpy arrays) following SURVEY.md section 8(d)."""
    rs = np.random.RandomState(seed)
    R = frames * boxes
    feats = np.zeros((bs, K, R, F), np.float32)
    boxes_a = np.zeros((bs, K, R, 12), np.float32)
    masks = np.zeros((bs, K, R), np.int64)
    tokens = np.zeros((bs, K, T), np.int64)
    targets = np.full((bs, K, T), -1, np.int64)

    def one_path():
        f = np.maximum(rs.standard_normal((frames, boxes, F)).astype(np.float32), 0.0)   # BUTD features are post-ReLU
        xy = np.sort(rs.uniform(0, 1, (frames, boxes, 2, 2)).astype(np.float32), axis=-1)
        b = np.ones((frames, boxes, 12), np.float32)
        b[..., 0], b[..., 2] = xy[..., 0, 0], xy[..., 0, 1]
        b[..., 1], b[..., 3] = xy[..., 1, 0], xy[..., 1, 1]
        b[..., 4] = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
        if finetune_heading:
            h = rs.uniform(-np.pi, np.pi, (frames, 1, 3)).astype(np.float32)
            b[..., 5], b[..., 6] = np.sin(h[..., 0]), np.cos(h[..., 0])
            b[..., 7], b[..., 8] = np.sin(h[..., 1]), np.cos(h[..., 1])
            b[..., 9], b[..., 10] = np.sin(h[..., 2]), np.cos(h[..., 2])
        nb = rs.randint(min(10, boxes), boxes + 1, size=frames)
        m = (np.arange(boxes)[None, :] < nb[:, None]).astype(np.int64)
        L = rs.randint(min(4, frames), frames + 1)
        m[L:] = 0
        # region 0 of each frame: mean feature, whole-image box (features_reader.py:170-177)
        f[:, 0] = (f * m[..., None]).sum(1) / np.maximum(m.sum(1, keepdims=True), 1)
        b[:, 0, :5] = np.array([0, 0, 1, 1, 1], np.float32)
        return f, b, m, L

    def one_instr():
        n = rs.randint(max(min_len, min(20, T - 2)), T - 1) if T > 3 else 1
        t = np.zeros(T, np.int64)
        t[0] = 101 if vocab > 103 else 1
        t[1:1 + n] = rs.randint(min(1000, vocab // 2), vocab, size=n)
        t[1 + n] = 102 if vocab > 103 else 2
        return t

    for i in range(bs):
        f0, b0, m0, L0 = one_path()
        ins = [one_instr() for _ in range(3)]
        for k in range(K):
            f, b, m = f0, b0, m0
            if k in (3, 4) and frames > 1:                      # frame permutations of the positive path
                perm = np.concatenate([rs.permutation(L0), np.arange(L0, frames)])
                f, b, m = f0[perm], b0[perm], m0[perm]
            elif k >= 5:                                         # replace some frames with fresh draws
                f1, b1, m1, _ = one_path()
                sw = rs.rand(frames) < 0.5
                sw[0] = False
                f = np.where(sw[:, None, None], f1, f0)
                b = np.where(sw[:, None, None], b1, b0)
                m = np.where(sw[:, None], m1, m0)
            b = b.copy()
            b[..., 11] = np.arange(frames, dtype=np.float32)[:, None]   # frame index stored as float (all_dataset.py:314)
            feats[i, k], boxes_a[i, k], masks[i, k] = f.reshape(R, F), b.reshape(R, 12), m.reshape(R)
            tokens[i, k] = ins[k] if k in (1, 2) else ins[0]

    instr_mask = tokens > 0
    # masked-language modelling (common.py:213-270): 15% of valid tokens, 80/10/10 mask/random/keep
    sel = (rs.rand(bs, K, T) < 0.15) & instr_mask
    sel[..., 0] = False
    targets[sel] = tokens[sel]
    r = rs.rand(bs, K, T)
    mask_id = 103 if vocab > 103 else 3
    tokens = np.where(sel & (r < 0.8), mask_id, tokens)
    rnd = rs.randint(min(1000, vocab // 2), vocab, size=(bs, K, T))
    tokens = np.where(sel & (r >= 0.8) & (r < 0.9), rnd, tokens)

    # masked-vision (common.py:272-300): 15% of valid regions; 90% of those get zeroed features
    tmask = ((rs.rand(bs, K, R) < 0.15) & (masks > 0)).astype(np.int64)
    logits = (rs.standard_normal((bs, K, R, C)) * 3.0).astype(np.float32)
    logits -= logits.max(-1, keepdims=True)
    sm = np.exp(logits)
    sm /= sm.sum(-1, keepdims=True)
    img_targets = np.where(tmask[..., None] > 0, sm, np.float32(1.0 / C)).astype(np.float32)
    zero = (tmask > 0) & (rs.rand(bs, K, R) < 0.9)
    feats[zero] = 0.0

    rank_target = np.zeros(bs, np.int64)
    rank_target[rs.rand(bs) < ignore_rank_frac] = -1
    opt_mask = np.ones((bs, K), bool)
    for j in range(opt_holes):
        opt_mask[(j + 1) % bs, K - 1 - (j % K)] = False

    return [rank_target, feats, boxes_a, masks, img_targets, tmask, tokens, instr_mask, targets,
            np.zeros((bs, K, 0), np.float32), np.zeros((bs, K, T), np.int64), np.zeros((bs, 2, R, T), np.int64),
            np.arange(bs, dtype=np.int64), opt_mask, np.zeros(bs, np.int64), np.zeros(bs, np.int64)]


def make_pool(bs: int, K: int, T: int, frames: int, boxes: int, F: int = 2048, C: int = 1601, vocab: int = 30522, seed: int = 1234):
    """The same kind of batch in the COMPACT form consumed by ytvln.batch (on-device assembly): every distinct frame once
    (`pool_*`, one entry per frame) + `index [bs, K, frames]` saying which pool frame each option shows at each position (-1 = padding)
    + un-masked tokens.  Option structure as in make_batch: 0 positive, 1-2 caption negatives sharing its frames, 3-4 frame
    permutations, 5-6 some frames swapped for other photos.  Returns (pool_features, pool_boxes, pool_probs, pool_masks, index,
    tokens [bs,K,T], instr_mask [bs,K,T])."""
    rs = np.random.RandomState(seed)
    per_item = frames + 2 * (frames // 2)                     # the path + replacement photos for the two "random" negatives
    P = bs * per_item
    pf = np.maximum(rs.standard_normal((P, boxes, F)).astype(np.float32), 0.0)
    pb = np.ones((P, boxes, 12), np.float32)
    xy = np.sort(rs.uniform(0, 1, (P, boxes, 2, 2)).astype(np.float32), axis=-1)
    pb[..., 0], pb[..., 2], pb[..., 1], pb[..., 3] = xy[..., 0, 0], xy[..., 0, 1], xy[..., 1, 0], xy[..., 1, 1]
    pb[..., 4] = (pb[..., 2] - pb[..., 0]) * (pb[..., 3] - pb[..., 1])
    nb = rs.randint(min(10, boxes), boxes + 1, size=P)
    pm = (np.arange(boxes)[None, :] < nb[:, None]).astype(np.int64)
    pf[:, 0] = (pf * pm[..., None]).sum(1) / np.maximum(pm.sum(1, keepdims=True), 1)
    pb[:, 0, :5] = np.array([0, 0, 1, 1, 1], np.float32)
    logits = (rs.standard_normal((P, boxes, C)) * 3.0).astype(np.float32)
    logits -= logits.max(-1, keepdims=True)
    pp = np.exp(logits)
    pp /= pp.sum(-1, keepdims=True)
    index = np.full((bs, K, frames), -1, np.int64)
    tokens = np.zeros((bs, K, T), np.int64)
    for i in range(bs):
        base = i * per_item
        L = rs.randint(min(4, frames), frames + 1)
        path = base + np.arange(L)
        spare = base + frames + np.arange(2 * (frames // 2))
        ins = []
        for _ in range(3):
            n = rs.randint(min(20, T - 2), T - 1)
            t = np.zeros(T, np.int64)
            t[0], t[1:1 + n], t[1 + n] = 101, rs.randint(1000, vocab, size=n), 102
            ins.append(t)
        for k in range(K):
            row = path.copy()
            if k in (3, 4):
                row = path[rs.permutation(L)]
            elif k >= 5:
                sw = np.nonzero(rs.rand(L) < 0.5)[0]
                sw = sw[sw > 0][:frames // 2]
                row[sw] = spare[(k - 5) * (frames // 2):(k - 5) * (frames // 2) + len(sw)]
            index[i, k, :L] = row
            tokens[i, k] = ins[k] if k in (1, 2) else ins[0]
    return pf, pb, pp.astype(np.float32), pm, index, tokens, (tokens > 0).astype(np.int64)


def to_torch(batch: List[np.ndarray], device="cpu"):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in batch]
