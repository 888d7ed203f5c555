"""GPU: host-side behaviour of the bf16-resident path (BASELINE configs[4]) around the kernels -- the bf16 weight copies follow every way a
parameter can change, gradient accumulation, checkpoints, eval mode, and the other loss combinations.  Reference point in every test: the SAME
path run from a clean state (bit-exact), or the fp32 path within the bf16 tolerance of tests/test_model_gpu.py (losses 2e-2)."""
import numpy as np
import pytest
import torch

from helpers import ZERO_DROP, args_ns
from test_model_gpu import build_lily, losses_of

pytestmark = pytest.mark.gpu
CFG = "tiny_2_2_1.json"          # 2 + 2 + 1 layers, hidden 256 / 4 heads: head dimension 64


def _batch(dev, seed=22, **kw):
    from ytvln import synth
    return synth.to_torch(synth.make_batch(**{**dict(bs=2, K=7, T=16, frames=2, boxes=4, seed=seed, ignore_rank_frac=0.0), **kw}), dev)


@pytest.fixture
def bf16_mode():
    from ytvln import ops
    ops.set_matmul_precision("bf16")
    yield
    ops.set_matmul_precision("fp32")


def _flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def _train(model, args, batch, steps, accumulate=1):
    from ytvln import utils_init as U
    from ytvln.vilbert_init import get_optimization
    args.gradient_accumulation_steps = accumulate
    opt, sched, _, _ = get_optimization(args, model, 20, None)
    for i in range(steps):
        U.train_step(model, opt, sched, batch, args, i, all_options=True)
    return opt, sched


def test_weight_copies_follow_in_place_edits_and_load_state_dict(dev, lib, bf16_mode):
    """After the optimizer's arenas exist the GEMMs read the bf16 weight arena the AdamW kernel maintains.  A parameter changed behind the
    optimizer's back (in-place edit, load_state_dict) must be re-rounded before its next use: the forward then equals the forward of a fresh
    model built from the same values."""
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    model, _ = build_lily(dev, CFG, args, seed=5)
    model.train()
    batch = _batch(dev)
    _train(model, args, batch, 2)                      # arena + bf16 arena exist, both refreshed by AdamW
    with torch.no_grad():
        model.bert.encoder.layer[0].attention.self.query.weight.mul_(0.5)          # in-place edit
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        sd["bert.encoder.v_layer[1]".replace("[1]", ".1") + ".intermediate.dense.weight"].add_(0.01)
    model.load_state_dict(sd)                          # copies into the arena views, version counters move
    model.eval()
    with torch.no_grad():
        _, total, per = losses_of(model, batch, args)
    fresh, _ = build_lily(dev, CFG, args, seed=99)
    fresh.load_state_dict(sd)
    fresh.to(dev).eval()
    with torch.no_grad():
        _, total_f, per_f = losses_of(fresh, batch, args)
    assert float(total) == float(total_f), (float(total), float(total_f))
    for k in per:
        assert torch.equal(per[k], per_f[k]), k


def test_gradient_accumulation_matches_one_large_step(dev, lib, bf16_mode):
    """Two backward passes on the same batch with gradient_accumulation_steps = 2 (train_step divides each loss by 2): the accumulated gradient
    equals the single-pass gradient up to the rounding of the bf16 activation gradients, so the parameters after the optimizer step agree to
    1e-3 of the update; and the run is reproducible bit for bit."""
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = _batch(dev)
    res = []
    for accumulate, steps in ((1, 1), (2, 2), (2, 2)):
        model, _ = build_lily(dev, CFG, args, seed=5)
        model.train()
        _train(model, args, batch, steps, accumulate)
        res.append(_flat(model))
    assert np.array_equal(res[1], res[2]), "accumulation must be reproducible"
    upd = np.abs(res[0] - res[1]).max()
    assert upd < 2e-4, upd          # lr 1e-3 bounds an AdamW update by ~1e-3 per element: the two runs must take (nearly) the same one


def test_checkpoint_round_trip_resumes_bit_exactly(dev, lib, bf16_mode, tmp_path):
    """save_model after 2 steps, resume through get_optimization in a NEW model / optimizer, step once more: identical to the uninterrupted run
    (the bf16 arena is rebuilt from the loaded fp32 masters)."""
    from ytvln import ops, utils_init as U
    from ytvln.vilbert_init import get_optimization
    args = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args.learning_rate = 1e-3
    batch = _batch(dev)
    model, _ = build_lily(dev, CFG, args, seed=5)
    model.train()
    opt, sched = _train(model, args, batch, 2)
    U.save_model(str(tmp_path), "ck", None, model, opt, sched, 0)
    U.train_step(model, opt, sched, batch, args, 2, all_options=True)
    want = _flat(model)
    args2 = args_ns(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)
    args2.learning_rate, args2.resume, args2.gradient_accumulation_steps = 1e-3, True, 1
    args2.from_pretrained = U.get_model_path(str(tmp_path), "ck")
    args2.save_name = "ck"
    args2.output_dir = str(tmp_path)
    model2, _ = build_lily(dev, CFG, args2, seed=77)
    model2.train()
    ck = torch.load(U.get_model_path(str(tmp_path), "ck"), map_location="cpu", weights_only=False)
    model2.load_state_dict(ck["model_state_dict"])
    opt2, sched2, _, _ = get_optimization(args2, model2, 20, None)
    opt2.load_state_dict(ck["optimizer_state_dict"])
    sched2.load_state_dict(ck["scheduler_state_dict"])
    U.train_step(model2, opt2, sched2, batch, args2, 2, all_options=True)
    got = _flat(model2)
    assert np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("flags", [dict(masked_language=True), dict(ranking=True), dict(masked_vision=True, traj_judge=True)])
def test_loss_subsets_and_eval_mode_against_fp32(dev, lib, flags):
    """Every loss combination trains in bf16 mode (tensors the reference leaves without a gradient stay None), and the eval-mode losses agree
    with the fp32 path within the bf16 tolerance."""
    from ytvln import ops
    args = args_ns(**flags)
    batch = _batch(dev, seed=31)
    model, _ = build_lily(dev, CFG, args, seed=6)
    model.eval()
    with torch.no_grad():
        _, total32, per32 = losses_of(model, batch, args)
    model.train()
    _, t, _ = losses_of(model, batch, args)
    t.backward()
    unused32 = {n for n, p in model.named_parameters() if p.grad is None}
    model.zero_grad(set_to_none=True)
    ops.set_matmul_precision("bf16")
    try:
        model.eval()
        with torch.no_grad():
            _, total16, per16 = losses_of(model, batch, args)
        model.train()
        _, t16, _ = losses_of(model, batch, args)
        t16.backward()
        unused16 = {n for n, p in model.named_parameters() if p.grad is None}
    finally:
        ops.set_matmul_precision("fp32")
    assert unused16 == unused32
    for k in per32:
        if not k.startswith("correct_"):
            a, b = float(per32[k]), float(per16[k])
            assert abs(a - b) <= 2e-2 * max(abs(a), 1e-6), (k, a, b)
    assert float(total16) != float(total32), "bf16 mode reproduced the fp32 loss exactly: the bf16 path did not run"
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), n


def test_g12_cfg4_finetune_full_size_bf16(dev, lib):
    """BASELINE configs[3] shapes on the bf16-resident path: fine-tune --ranking, 96 rows, R = 7 x 36 = 252 regions -- NOT a multiple of the
    32-row attention tiles (clamped tail rows in every bf16 attention kernel) -- against the reference golden with the bf16 bars."""
    from ytvln import synth
    from test_model_gpu import FULL_CFG, _bf16_check, gold
    g = gold("g12_cfg4_full_n96.npz")
    args = args_ns(ranking=True, pretrain=False, num_negatives=2)
    model, _ = build_lily(dev, FULL_CFG, args, seed=33)
    batch = synth.to_torch(synth.make_batch(bs=16, K=6, T=80, frames=7, boxes=36, seed=43, finetune_heading=True), dev)
    _bf16_check(model, batch, args, g)


def test_inference_rerank_shapes_bf16_against_fp32(dev, lib):
    """The re-ranking inference shapes (30 beams, R = 8 x 101 = 808 regions, T = 60, a ragged opt_mask, eval mode) through eval_epoch: the bf16
    scores stay within 2e-2 of the fp32 scores' spread and both pick the same best beam wherever the fp32 margin exceeds that."""
    from ytvln import ops, synth, utils_init as U
    args = args_ns(ranking=True, pretrain=False)
    model, _ = build_lily(dev, CFG, args, seed=31)
    model.eval()
    nb = synth.make_batch(bs=2, K=30, T=60, frames=8, boxes=101, seed=77, finetune_heading=True, ignore_rank_frac=0.0)
    nb[13][1, 25:] = False
    nb[12] = np.array([[4051, 0], [4051, 2]], np.int64)
    batch_cpu = synth.to_torch(nb)
    s32 = U.eval_epoch(model, [batch_cpu], args)
    ops.set_matmul_precision("bf16")
    try:
        s16 = U.eval_epoch(model, [batch_cpu], args)
    finally:
        ops.set_matmul_precision("fp32")
    assert [s[0] for s in s16] == [s[0] for s in s32]
    a, b = torch.tensor([s[1] for s in s32]), torch.tensor([s[1] for s in s16])
    valid = batch_cpu[13]
    spread = float(a[valid].max() - a[valid].min())
    assert float((a - b)[valid].abs().max()) < 2e-2 * max(spread, 1.0), (float((a - b)[valid].abs().max()), spread)
    assert not torch.equal(a[valid], b[valid]), "bf16 mode reproduced the fp32 scores exactly: the bf16 path did not run"
    for ra, rb, v in zip(a, b, valid):
        top2 = torch.topk(ra[v], 2).values
        if float(top2[0] - top2[1]) > 4e-2 * max(spread, 1.0):
            assert int(torch.argmax(ra[v])) == int(torch.argmax(rb[v]))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_full_model_backward_is_bit_reproducible(dev, lib, precision):
    """The FULL model at BASELINE configs[3] size (96 rows, R = 252, T = 80: ~20 000 one-wave attention workgroups per layer), forward +
    backward three times on the same weights and batch, dropout off: losses and EVERY gradient must come out bit-identical (all reductions
    have a fixed order; two HIP streams do not change that).  This is the test that would have caught the write-after-read race on the LDS
    tile buffers of the bf16 attention kernels, which the golden's 5 % gradient-norm bars let through five times in six."""
    from ytvln import ops, synth
    from test_model_gpu import FULL_CFG
    args = args_ns(ranking=True, pretrain=False, num_negatives=2)
    model, _ = build_lily(dev, FULL_CFG, args, seed=33)
    batch = synth.to_torch(synth.make_batch(bs=16, K=6, T=80, frames=7, boxes=36, seed=43, finetune_heading=True), dev)
    model.train()
    ops.set_matmul_precision(precision)
    runs = []
    try:
        for i in range(3):
            model.zero_grad(set_to_none=True)
            junk = torch.full((96 * 252, 1024), 3.0, device=dev)      # fresh activations do not land on the last run's values
            del junk
            _, total, _ = losses_of(model, batch, args)
            total.backward()
            torch.cuda.synchronize()
            runs.append((float(total), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        ops.set_matmul_precision("fp32")
    for i in (1, 2):
        assert runs[i][0] == runs[0][0], (i, runs[i][0], runs[0][0])
        bad = [(n, float((g - runs[0][1][n]).abs().max())) for n, g in runs[i][1].items() if not torch.equal(g, runs[0][1][n])]
        assert not bad, (i, bad[:6])
