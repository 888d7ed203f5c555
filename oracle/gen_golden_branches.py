"""Goldens for the three switches that are OFF in every target config but exist on the path (VERDICT r1 "missing" 3-4): `in_batch_pairs`
and `fast_mode` of BertEncoder (vilbert/vilbert.py:771-782) and the `predict_feature` MSE branch of BertForMultiModalPreTraining
(:1391, 1430-1434).  TEST INFRASTRUCTURE ONLY; build container only: runs the REAL reference on the tiny config and stores what it returns.

    python oracle/gen_golden_branches.py        -> tests/golden/g13_branches.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_golden import GOLD, ZERO_DROP, load_cfg, np_  # noqa: E402
from ytvln import synth  # noqa: E402


def model_inputs(nb, K=0):
    b = synth.to_torch(nb)
    return b[6][:, K], b[1][:, K], b[2][:, K], b[10][:, K], b[7][:, K], b[3][:, K]      # ids, feat, loc, segment, text mask, image mask


def bert_model(R, seed, **over):
    rcfg, _ = load_cfg(R, "tiny_2_2_1.json", **ZERO_DROP, **over)
    m = R.vilbert.BertModel(rcfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    W = synth.make_weights(shapes, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()})
    return m.train()


def run_bert(m, ids, feat, loc, seg, tmask, vmask, out, tag):
    seq_t, seq_v, pool_t, pool_v, _ = m(ids, feat, loc, seg, tmask, vmask)
    loss = (pool_t * pool_v).sum() + 0.01 * seq_t.sum() + 0.01 * seq_v.sum()
    loss.backward()
    out[tag + "/seq_t"], out[tag + "/seq_v"], out[tag + "/pool_t"], out[tag + "/pool_v"] = np_(seq_t), np_(seq_v), np_(pool_t), np_(pool_v)
    out[tag + "/loss"] = np_(loss)
    names, gn = [], []
    for n, p in m.named_parameters():
        if p.grad is not None:
            names.append(n); gn.append(p.grad.double().norm().item())
    out[tag + "/grad_names"], out[tag + "/grad_norms"] = np.array(names), np.array(gn)


def main():
    R = ref_import.import_reference()
    torch.set_num_threads(8)
    out = {}
    # in_batch_pairs: 3 texts x 3 images -> 9 rows
    nb = synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=51)
    run_bert(bert_model(R, 21, in_batch_pairs=True), *model_inputs(nb), out, "pairs")
    # fast_mode: 1 text against 4 images
    nb4 = synth.make_batch(bs=4, K=1, T=12, frames=2, boxes=5, seed=52)
    ids, feat, loc, seg, tmask, vmask = model_inputs(nb4)
    run_bert(bert_model(R, 22, fast_mode=True), ids[:1], feat, loc, seg[:1], tmask[:1], vmask, out, "fast")
    # predict_feature: MSE over the masked regions
    rcfg, _ = load_cfg(R, "tiny_2_2_1.json", **ZERO_DROP, predict_feature=True)
    m = R.vilbert.BertForMultiModalPreTraining(rcfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, 23).items()})
    m.eval()          # (BertPreTrainingHeads carries a hard-wired Dropout(0.1) in front of the NSP head: eval mode keeps torch's RNG out of the fixture)
    nb3 = synth.make_batch(bs=3, K=1, T=12, frames=2, boxes=5, seed=53)
    b = synth.to_torch(nb3)
    ids, feat, loc, vmask = b[6][:, 0], b[1][:, 0], b[2][:, 0], b[3][:, 0]
    imask, labels = b[7][:, 0], b[8][:, 0]
    img_label = b[5][:, 0, 1:]
    g = torch.Generator().manual_seed(7)
    img_target = torch.randn(b[4][:, 0, 1:].shape, generator=g)
    nsl = torch.tensor([0, 1, 0])
    l = m(ids, feat, loc, None, imask, vmask, labels, img_label, img_target, nsl)
    (l[0] + l[1] + l[2]).sum().backward()
    out["mse/losses"] = np.array([float(x) for x in l])
    out["mse/img_target"], out["mse/nsl"] = img_target.numpy(), nsl.numpy()
    names, gn = [], []
    for n, p in m.named_parameters():
        if p.grad is not None:
            names.append(n); gn.append(p.grad.double().norm().item())
    out["mse/grad_names"], out["mse/grad_norms"] = np.array(names), np.array(gn)
    np.savez_compressed(os.path.join(GOLD, "g13_branches.npz"), **out)
    print("g13 ok", float(out["pairs/loss"]), float(out["fast/loss"]), out["mse/losses"])


if __name__ == "__main__":
    main()
