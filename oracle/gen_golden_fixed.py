"""Golden for the `fixed_t_layer` / `fixed_v_layer` switches of BertEncoder (vilbert/vilbert.py:742-764: the first layers of each stream run
under no_grad, their parameters -- and everything below them -- get no gradient).  TEST INFRASTRUCTURE ONLY; build container only: runs the
REAL reference and stores what it returns.

The tiny config is deepened so that both switches bite: 4 text / 3 image layers, co-attention after (t 2, v 1) and (t 3, v 2),
fixed_t_layer = 2, fixed_v_layer = 1.

    python oracle/gen_golden_fixed.py        -> tests/golden/g17_fixed_layers.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_golden import GOLD  # noqa: E402
from gen_golden_branches import bert_model, model_inputs, run_bert  # noqa: E402
from ytvln import synth  # noqa: E402

DEEP = dict(num_hidden_layers=4, v_num_hidden_layers=3, t_biattention_id=[2, 3], v_biattention_id=[1, 2])
RECIPE = dict(bs=3, K=1, T=12, frames=2, boxes=5, seed=61)


def main():
    R = ref_import.import_reference()
    torch.set_num_threads(8)
    out = {}
    nb = synth.make_batch(**RECIPE)
    run_bert(bert_model(R, 31, fixed_t_layer=2, fixed_v_layer=1, **DEEP), *model_inputs(nb), out, "fixed")
    run_bert(bert_model(R, 31, **DEEP), *model_inputs(nb), out, "free")          # the same model without the switches: same outputs, more gradients
    assert np.array_equal(out["fixed/seq_t"], out["free/seq_t"]) and np.array_equal(out["fixed/pool_v"], out["free/pool_v"])
    frozen = set(out["free/grad_names"].tolist()) - set(out["fixed/grad_names"].tolist())
    assert any(n.startswith("encoder.layer.1.") for n in frozen) and any(n.startswith("encoder.v_layer.0.") for n in frozen)
    assert any(n.startswith("embeddings.") for n in frozen) and not any(n.startswith("encoder.layer.2.") for n in frozen)
    np.savez_compressed(os.path.join(GOLD, "g17_fixed_layers.npz"), **out)
    print("g17 ok", float(out["fixed/loss"]), len(out["fixed/grad_names"]), "of", len(out["free/grad_names"]), "parameters receive a gradient")


if __name__ == "__main__":
    main()
