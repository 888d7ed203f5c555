"""The kernels of the bf16-resident path (BASELINE configs[4]; include/ytvln.h: ytvln_gemm_bf16, ytvln_attn_*_bf16, ytvln_ln_*_bf16, the bf16
loss gradients and the AdamW bf16 copy) against fp64 restatements on the SAME bf16-rounded inputs.

Tolerances (stated per test): products of bf16 values are exact in fp32 and accumulate in fp32, so an fp32 result carries fp32 accumulation
noise only (3e-6 relative to the largest entry); a bf16 result adds ONE rounding (2^-9 = 2e-3 relative per element).  Attention rounds P and dS
to bf16 on the way into the matrix instruction: relative L2 <= 1e-2 forward, <= 2e-2 for dQ / dK / dV -- the bars the round-2 bf16-operand
kernels were held to."""
import ctypes
import math

import pytest
import torch

from helpers import rel_l2
from test_kernels_gpu import rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _gemm(ops, A, ta, B, tb, C, M, N, K, **kw):
    return ops._gemm_bf16(A, A.stride(0), ta, B, B.stride(0), tb, C, C.stride(0), M, N, K, **kw)


def _relmax(got, ref):
    return float((got.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (130, 72, 200), (4480, 768, 768), (1000, 520, 1088), (300, 1608, 96), (257, 264, 72),
                                   (1024, 1024, 16128), (768, 3072, 4480), (8, 8, 8), (40, 24, 1000), (57, 1001, 64), (33, 17, 12)])
def test_gemm_bf16_all_layouts(dev, lib, M, N, K):
    """Every operand layout (contraction contiguous / k-major through ds_read_b64_tr_b16), both output types, ragged M / N / K (register-staged
    K tail), split-K shapes, and shapes only the generic kernel takes (unaligned); row sums of a k-major A ride on the launch."""
    from ytvln import ops
    for ta in (0, 1):
        for tb in (0, 1):
            g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
            A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev).to(BF)
            B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev).to(BF)
            bias = torch.randn(N, generator=g).to(dev)
            ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
            for cdt, tol in ((torch.float32, 3e-6), (BF, 6e-3)):
                C = torch.full((M, N), float("nan"), device=dev, dtype=cdt)
                rs = torch.full((M,), float("nan"), device=dev)
                done = _gemm(ops, A, ta, B, tb, C, M, N, K, bias=bias, rowsum=rs if ta else None)
                assert bool(torch.isfinite(C.float()).all())
                assert _relmax(C, ref) < tol * max(1.0, math.sqrt(K / 1024)), (ta, tb, cdt, _relmax(C, ref))
                if ta:
                    assert done and _relmax(rs, A.double().sum(0)) < 1e-5


def test_gemm_bf16_epilogues_splitk_and_padding(dev, lib):
    from ytvln import ops
    from ytvln._lib import GEMM_A_ZERO_PADDED
    gelu = lambda x: x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))                                            # noqa: E731
    dgelu = lambda x: 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)   # noqa: E731
    M, N, K = 520, 392, 256
    g = torch.Generator().manual_seed(5)
    A, B, bias = torch.randn(M, K, generator=g).to(dev).to(BF), torch.randn(N, K, generator=g).to(dev).to(BF), torch.randn(N, generator=g).to(dev)
    lin = A.double() @ B.double().t()
    pre = lin + bias.double()
    for cdt, tol in ((torch.float32, 3e-6), (BF, 6e-3)):
        C, aux = torch.empty(M, N, device=dev, dtype=cdt), torch.empty(M, N, device=dev, dtype=BF)
        _gemm(ops, A, 0, B, 1, C, M, N, K, bias=bias, aux=aux, ldaux=N, epi=1)
        assert _relmax(C, gelu(pre)) < tol and _relmax(aux, pre) < 6e-3
        _gemm(ops, A, 0, B, 1, C, M, N, K, bias=bias, epi=2)
        assert _relmax(C, torch.relu(pre)) < tol
        z = torch.randn(M, N, generator=g).to(dev).to(BF)
        _gemm(ops, A, 0, B, 1, C, M, N, K, aux=z, ldaux=N, epi=3)
        assert _relmax(C, lin * dgelu(z.double())) < tol
        _gemm(ops, A, 0, B, 1, C, M, N, K, aux=z, ldaux=N, epi=4)
        assert _relmax(C, lin * (z.double() > 0)) < tol
        C0 = torch.randn(M, N, generator=g).to(dev).to(cdt)
        C = C0.clone()
        _gemm(ops, A, 0, B, 1, C, M, N, K, beta=1.0)
        assert _relmax(C, lin + C0.double()) < tol
    # deterministic split-K (weight-gradient shape) accumulating into an existing C
    M, N, K = 1024, 1024, 16128
    A, B = torch.randn(K, M, generator=g).to(dev).to(BF), torch.randn(K, N, generator=g).to(dev).to(BF)
    for cdt, tol in ((torch.float32, 2e-5), (BF, 6e-3)):
        C0 = torch.randn(M, N, generator=g).to(dev).to(cdt)
        C, C2 = C0.clone(), C0.clone()
        _gemm(ops, A, 1, B, 0, C, M, N, K, beta=1.0)
        _gemm(ops, A, 1, B, 0, C2, M, N, K, beta=1.0)
        assert torch.equal(C, C2), "split-K must be deterministic"
        assert _relmax(C, A.double().t() @ B.double() + C0.double()) < tol
    # zero-padded A (the 1601- / 30522-wide logit gradients): contraction-contiguous with K % 8 != 0, and k-major with M % 8 != 0
    M, N, K = 300, 768, 1601
    ld = (K + 63) // 64 * 64
    Afull = torch.zeros(M, ld, device=dev, dtype=BF)
    Afull[:, :K] = torch.randn(M, K, generator=g).to(dev).to(BF)
    B = torch.randn(K, N, generator=g).to(dev).to(BF)
    C = torch.empty(M, N, device=dev, dtype=BF)
    ops._gemm_bf16(Afull, ld, 0, B, N, 0, C, N, M, N, K, flags=GEMM_A_ZERO_PADDED)
    assert _relmax(C, Afull[:, :K].double() @ B.double()) < 6e-3
    Kc = 520
    A2 = torch.zeros(Kc, ld, device=dev, dtype=BF)
    A2[:, :K] = torch.randn(Kc, K, generator=g).to(dev).to(BF)
    X = torch.randn(Kc, N, generator=g).to(dev).to(BF)
    C, rs = torch.empty(K, N, device=dev), torch.empty(K, device=dev)
    assert ops._gemm_bf16(A2, ld, 1, X, N, 0, C, N, K, N, Kc, flags=GEMM_A_ZERO_PADDED, rowsum=rs)
    assert _relmax(C, A2[:, :K].double().t() @ X.double()) < 3e-6 and _relmax(rs, A2[:, :K].double().sum(0)) < 1e-5


@pytest.mark.parametrize("form", [1, 2, 3, 4])
def test_gemm_bf16_main_loop_forms_agree_with_the_shipped_one(dev, lib, form):
    """The opt-in main loops of the bf16 GEMM (run-time option GEMM_BF16_FORM: DMA issue inside the matrix phases, 32-deep tiles in five-slot rings,
    four waves of 128x128) accumulate every output element in the same k order as the shipped form (form 3: up to fp32 rounding): bit-identical results on both operand
    layouts a bf16 C takes, ragged K / M / N included (form 4 hands shapes it has no instantiation for back to the shipped kernel)."""
    from ytvln import _lib, ops
    for M, N, K, tb in ((2048, 1024, 1024, 1), (2048, 1024, 1024, 0), (1500, 520, 1088, 1), (4480, 768, 3072, 0), (512, 512, 200, 1)):
        g = torch.Generator().manual_seed(M + N + K + tb)
        A = torch.randn(M, K, generator=g).to(dev).to(BF)
        B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev).to(BF)
        bias = torch.randn(N, generator=g).to(dev)
        C2_init = torch.randn(M, N, generator=g).to(dev).to(BF)
        outs = []
        for f in (0, form):
            prev = _lib.set_option("GEMM_BF16_FORM", f)
            try:
                C = torch.full((M, N), float("nan"), device=dev, dtype=BF)
                aux = torch.empty(M, N, device=dev, dtype=BF)
                _gemm(ops, A, 0, B, tb, C, M, N, K, bias=bias, aux=aux, ldaux=N, epi=1)
                C2 = C2_init.clone()
                _gemm(ops, A, 0, B, tb, C2, M, N, K, beta=1.0)
                torch.cuda.synchronize()
                outs.append((C, aux, C2))
            finally:
                _lib.set_option("GEMM_BF16_FORM", prev)
        for x, y in zip(*outs):
            if form == 3:      # 32-deep tiles group the k indices of a matrix instruction differently: same sums up to fp32 rounding, one bf16 ulp at most
                assert float((x.float() - y.float()).abs().max()) <= 2 ** -7 * float(y.float().abs().max()), (form, M, N, K, tb)
            else:
                assert torch.equal(x, y), (form, M, N, K, tb, float((x.float() - y.float()).abs().max()))


def test_gemm_bf16_wide_stores_are_the_same_bits(dev, lib):
    """GEMM_BF16_WIDE: interior bf16 output tiles leave through an LDS transpose and 16-byte stores instead of 2-byte stores from the accumulator
    layout -- the same values rounded once: bit-identical C (and saved pre-activation) with the option on and off, for both tile sizes, every
    epilogue (the GELU one keeps the 2-byte path under either setting), ragged edges (edge tiles keep the narrow path) and a C with a leading dimension that is not a multiple of 8 (falls back)."""
    from ytvln import _lib, ops
    for M, N, K, tb, ldc in ((4480, 768, 768, 1, 768), (2048, 1024, 1024, 0, 1024), (1500, 520, 1088, 1, 520), (300, 200, 64, 1, 200), (512, 256, 128, 1, 260)):
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(dev).to(BF)
        B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev).to(BF)
        bias = torch.randn(N, generator=g).to(dev)
        outs = []
        for w in (0, 1):
            prev = _lib.set_option("GEMM_BF16_WIDE", w)
            try:
                res = []
                for epi in (0, 1, 2):
                    C = torch.full((M, ldc), 3.0, device=dev, dtype=BF)
                    aux = torch.full((M, N), 5.0, device=dev, dtype=BF) if epi == 1 else None
                    ops._gemm_bf16(A, K, 0, B, B.stride(0), tb, C, ldc, M, N, K, bias=bias, aux=aux, ldaux=N if aux is not None else 0, epi=epi)
                    torch.cuda.synchronize()
                    res += [C] + ([aux] if aux is not None else [])
                outs.append(res)
            finally:
                _lib.set_option("GEMM_BF16_WIDE", prev)
        for x, y in zip(*outs):
            assert torch.equal(x, y), (M, N, K, tb, ldc, float((x.float() - y.float()).abs().max()))
        assert float(outs[1][0][:, N:].float().sub(3.0).abs().max()) == 0 if ldc > N else True          # padding columns untouched


def _ref_attention(q, k, v, mask, heads, keep=None, p=0.0):
    N, Tq, H = q.shape
    d = H // heads
    qh, kh, vh = (t.view(N, -1, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(d) + mask[:, None, None, :]
    pr = torch.softmax(s, -1)
    pd = pr if keep is None else pr * keep / (1 - p)
    return (pd @ vh).permute(0, 2, 1, 3).reshape(N, Tq, H)


@pytest.mark.parametrize("N,heads,d,Tq,Tk", [(2, 8, 128, 288, 288), (2, 8, 128, 80, 288), (2, 8, 128, 288, 80), (3, 12, 64, 80, 80), (1, 8, 128, 576, 576),
                                             (2, 2, 128, 37, 101), (1, 1, 128, 1, 3), (2, 3, 64, 33, 65), (1, 2, 64, 100, 1000)])
def test_attention_bf16(dev, lib, N, heads, d, Tq, Tk):
    """Forward and backward of the bf16-resident attention kernels against fp64 on the same bf16 inputs: packed strided q | k | v as the fused
    projection writes them, ragged tiles, a masked tail and a fully masked row set."""
    from ytvln import ops
    H = heads * d
    A = rnd(dev, N * Tq, 3 * H, seed=1).to(BF)
    B = rnd(dev, N * Tk, 3 * H, seed=2).to(BF)
    mask = torch.zeros(N, Tk, device=dev)
    mask[0, Tk - max(1, Tk // 4):] = -10000.0
    if N > 1:
        mask[1, :] = -10000.0
    out = torch.empty(N * Tq, H, device=dev, dtype=BF)
    scale = 1 / math.sqrt(d)
    lse = ops._attn_fwd(A, 0, 3 * H, B, H, 3 * H, B, 2 * H, 3 * H, mask, out, N, heads, Tq, Tk, d, scale, 0.0, None, 0)
    qd = A[:, :H].double().view(N, Tq, H).requires_grad_(True)
    kd = B[:, H:2 * H].double().reshape(N, Tk, H).requires_grad_(True)
    vd = B[:, 2 * H:].double().reshape(N, Tk, H).requires_grad_(True)
    ref = _ref_attention(qd, kd, vd, mask.double(), heads)
    part = [n for n in range(N) if not bool((mask[n] != 0).all())]          # (fully masked rows: fp32 score quantisation at -10000, see test_attention_fwd_bwd)
    assert rel_l2(out.view(N, Tq, H)[part], ref[part]) < 1e-2
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all())
    dout = rnd(dev, N * Tq, H, seed=3).to(BF)
    ref.backward(dout.double().view(N, Tq, H))
    gA, gB = torch.zeros_like(A), torch.zeros_like(B)
    ops._attn_bwd(A, 0, 3 * H, B, H, 3 * H, B, 2 * H, 3 * H, mask, out, dout, lse, gA, 0, 3 * H, gB, H, 3 * H, gB, 2 * H, 3 * H, N, heads, Tq, Tk, d,
                  scale, 0.0, None, 0)
    assert rel_l2(gA[:, :H].reshape(N, Tq, H)[part], qd.grad[part]) < 2e-2, "dq"
    assert rel_l2(gB[:, H:2 * H].reshape(N, Tk, H)[part], kd.grad[part]) < 2e-2, "dk"
    assert rel_l2(gB[:, 2 * H:].reshape(N, Tk, H)[part], vd.grad[part]) < 2e-2, "dv"
    assert float(gA[:, H:].float().abs().max()) == 0 and float(gB[:, :H].float().abs().max()) == 0, "only the addressed column blocks are written"
    assert bool(torch.isfinite(gA.float()).all()) and bool(torch.isfinite(gB.float()).all())


@pytest.mark.parametrize("d", [128, 64])
def test_attention_bf16_dropout_and_pair(dev, lib, d):
    """The probability-dropout mask of the bf16 kernels is the fp32 kernels' (same per-score hash): recovered through V = identity columns, then
    forward / backward against fp64 with that mask; and the two directions of BertBiAttention in one launch equal two single launches bit for bit."""
    from ytvln import ops
    N, heads, Tq, p = 2, 2, 40, 0.2
    Tk = 96 if d >= 96 else 64
    H = heads * d
    st = ops.DropoutState(dev)
    site, scale = 5, 1 / math.sqrt(d)
    z, zk = torch.zeros(N * Tq, H, device=dev, dtype=BF), torch.zeros(N * Tk, H, device=dev, dtype=BF)
    eye = torch.zeros(N, Tk, heads, d, device=dev)
    for j in range(Tk):
        eye[:, j, :, j] = 1.0
    eye = eye.reshape(N * Tk, H).to(BF)
    out = torch.empty(N * Tq, H, device=dev, dtype=BF)
    ops._attn_fwd(z, 0, H, zk, 0, H, eye, 0, H, None, out, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
    keep = (out.float().view(N, Tq, heads, d)[..., :Tk] > 0).permute(0, 2, 1, 3).double()
    out32 = torch.empty(N * Tq, H, device=dev)
    ops._attn_fwd(z.float(), 0, H, zk.float(), 0, H, eye.float(), 0, H, None, out32, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
    assert torch.equal(keep, (out32.view(N, Tq, heads, d)[..., :Tk] > 0).permute(0, 2, 1, 3).double()), "bf16 and fp32 kernels must draw the same mask"
    assert abs(float(keep.mean()) - (1 - p)) < 0.02
    q, k, v = (rnd(dev, N * T, H, seed=sd).to(BF) for T, sd in ((Tq, 1), (Tk, 2), (Tk, 3)))
    mask = torch.zeros(N, Tk, device=dev)
    mask[0, Tk * 5 // 6:] = -10000.0
    lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
    qd, kd, vd = (t.double().view(N, -1, H).requires_grad_(True) for t in (q, k, v))
    ref = _ref_attention(qd, kd, vd, mask.double(), heads, keep, p)
    assert rel_l2(out.view(N, Tq, H), ref) < 1e-2
    dout = rnd(dev, N * Tq, H, seed=4).to(BF)
    ref.backward(dout.double().view(N, Tq, H))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, heads, Tq, Tk, d, scale, p, st.tensor, site)
    assert rel_l2(dq.view(N, Tq, H), qd.grad) < 2e-2 and rel_l2(dk.view(N, Tk, H), kd.grad) < 2e-2 and rel_l2(dv.view(N, Tk, H), vd.grad) < 2e-2
    # pair launch (text queries over regions | region queries over text) == two single launches
    R, T, Hb = 72, 20, H
    q1, kv1, q2, kv2 = rnd(dev, N * R, Hb, seed=11).to(BF), rnd(dev, N * R, 2 * Hb, seed=12).to(BF), rnd(dev, N * T, Hb, seed=13).to(BF), rnd(dev, N * T, 2 * Hb, seed=14).to(BF)
    m1, m2 = torch.zeros(N, R, device=dev), torch.zeros(N, T, device=dev)
    m1[1, R - 9:] = -10000.0
    c1, c2, l1, l2 = ops.CoAttentionFn.apply(q1, kv1, q2, kv2, m1, m2, N, R, T, heads, p, p, st.tensor, 7, 8)
    s1 = torch.empty_like(c1)
    ls1 = ops._attn_fwd(q2, 0, Hb, kv1, 0, 2 * Hb, kv1, Hb, 2 * Hb, m1, s1, N, heads, T, R, d, scale, p, st.tensor, 7)
    s2 = torch.empty_like(c2)
    ls2 = ops._attn_fwd(q1, 0, Hb, kv2, 0, 2 * Hb, kv2, Hb, 2 * Hb, m2, s2, N, heads, R, T, d, scale, p, st.tensor, 8)
    assert torch.equal(c1, s1) and torch.equal(c2, s2) and torch.equal(l1, ls1) and torch.equal(l2, ls2)


@pytest.mark.parametrize("rows,H", [(7, 32), (50, 48), (4480, 768), (2016, 1024), (5, 2048)])
def test_layernorm_bf16(dev, lib, rows, H):
    """bf16 rows in and out, fp32 arithmetic: y within one bf16 rounding of the fp64 LayerNorm of the same inputs; the input gradient within
    one rounding as well; gamma / beta gradients (fp32 partial sums) at fp32 accuracy given the rounded s."""
    from ytvln import ops
    x, res = rnd(dev, rows, H, seed=1).to(BF), rnd(dev, rows, H, seed=2).to(BF)
    gamma = (1 + 0.1 * rnd(dev, H, seed=3)).requires_grad_(True)
    beta = (0.1 * rnd(dev, H, seed=4)).requires_grad_(True)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    y = ops.add_layer_norm(xr, rr, gamma, beta, 1e-12)
    assert y.dtype == BF
    dy = rnd(dev, rows, H, seed=5).to(BF)
    y.backward(dy)
    xd, rd, gd, bd = x.double().requires_grad_(True), res.double().requires_grad_(True), gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    s = xd + rd
    yr = gd * (s - s.mean(-1, keepdim=True)) / torch.sqrt(s.var(-1, unbiased=False, keepdim=True) + 1e-12) + bd
    yr.backward(dy.double())
    assert rel_l2(y, yr) < 4e-3
    assert rel_l2(xr.grad, xd.grad) < 8e-3 and torch.equal(xr.grad, rr.grad)
    assert rel_l2(gamma.grad, gd.grad) < 5e-3 and rel_l2(beta.grad, bd.grad) < 1e-5


def test_linear_ffn_bf16_autograd_and_weight_copies(dev, lib):
    """ops.linear / ops.ffn_res on bf16 hidden states: outputs, input gradients (bf16) and weight / bias gradients (fp32) against fp64 autograd on
    the bf16-rounded operands; the fp32 `out_fp32` exit; and the cached bf16 weight copy follows in-place edits of the parameter."""
    from ytvln import ops
    M, K, I, N = 300, 256, 512, 192
    x = rnd(dev, M, K, seed=1).to(BF).requires_grad_(True)
    w1, b1 = (0.05 * rnd(dev, I, K, seed=2)).requires_grad_(True), (0.1 * rnd(dev, I, seed=3)).requires_grad_(True)
    w2, b2 = (0.05 * rnd(dev, N, I, seed=4)).requires_grad_(True), (0.1 * rnd(dev, N, seed=5)).requires_grad_(True)
    y, res = ops.ffn_res(x, w1, b1, w2, b2)
    assert y.dtype == BF and res is not None
    dy = rnd(dev, M, N, seed=6).to(BF)
    y.backward(dy)
    xd = x.detach().double().requires_grad_(True)
    w1d, w2d = w1.detach().to(BF).double().requires_grad_(True), w2.detach().to(BF).double().requires_grad_(True)
    b1d, b2d = b1.detach().double().requires_grad_(True), b2.detach().double().requires_grad_(True)
    h = torch.nn.functional.gelu(xd @ w1d.t() + b1d)
    yr = h @ w2d.t() + b2d
    yr.backward(dy.double())
    assert rel_l2(y, yr) < 6e-3 and rel_l2(x.grad, xd.grad) < 1e-2
    for got, ref in ((w1.grad, w1d.grad), (w2.grad, w2d.grad), (b1.grad, b1d.grad), (b2.grad, b2d.grad)):
        assert got.dtype == torch.float32 and rel_l2(got, ref) < 1e-2
    # single projection with activation, fp32 exit, strided input (first token of every row)
    hs = rnd(dev, 6, 5, K, seed=7).to(BF).requires_grad_(True)
    w, b = torch.nn.Parameter(0.05 * rnd(dev, N, K, seed=8)), torch.nn.Parameter(0.1 * rnd(dev, N, seed=9))
    out = ops.linear(hs[:, 0], w, b, "relu", out_fp32=True)
    assert out.dtype == torch.float32
    g = rnd(dev, 6, N, seed=10)
    out.backward(g)
    hd, wd, bdd = hs.detach().double().requires_grad_(True), w.detach().to(BF).double().requires_grad_(True), b.detach().double().requires_grad_(True)
    outr = torch.relu(hd[:, 0] @ wd.t() + bdd)
    outr.backward(g.to(BF).double())
    assert rel_l2(out, outr) < 1e-5 and rel_l2(hs.grad, hd.grad) < 1e-2 and rel_l2(w.grad, wd.grad) < 1e-2 and rel_l2(b.grad, bdd.grad) < 1e-2
    # the cached bf16 copy of a parameter outside any optimizer arena follows in-place edits (version counter)
    wb0 = ops._bf16_weight(w)
    assert torch.equal(wb0, w.detach().to(BF)) and ops._bf16_weight(w) is wb0
    with torch.no_grad():
        w.mul_(2.0)
    assert torch.equal(ops._bf16_weight(w), w.detach().to(BF))


def test_loss_gradients_bf16_and_adamw_copy(dev, lib):
    """bf16 logits (bf16-resident path): loss = the fp32 kernel's on the same (rounded) logits; gradient = that fp32 gradient rounded once to bf16,
    with ZERO padding up to the leading dimension.  The AdamW kernel with the bf16 copy updates the fp32 master exactly as the plain kernel and
    leaves bf16(p) beside it."""
    from ytvln import ops, _lib
    M, V, LD = 300, 1601, 1608
    buf = torch.full((M, LD), 7.0, device=dev, dtype=BF)          # junk in the padding columns of the logits must not matter
    buf[:, :V] = rnd(dev, M, V, seed=1).to(BF)
    logits_b = buf[:, :V].detach().requires_grad_(True)
    logits = logits_b.detach().float().requires_grad_(True)
    tgt = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2)).to(dev)
    tgt[::7] = -1
    l32 = ops.cross_entropy(logits, tgt, -1)
    l32.backward()
    lb = ops.cross_entropy(logits_b, tgt, -1)
    lb.backward()
    gb = logits_b.grad
    assert lb.dtype == torch.float32 and float(lb) == float(l32)
    assert gb.dtype == BF and gb.shape == (M, V) and torch.equal(gb, logits.grad.to(BF))
    if gb.stride(0) != V:          # (autograd may repack the gradient; when it keeps the kernel's buffer the padding must be zero)
        full = torch.as_strided(gb, (M, gb.stride(0)), (gb.stride(0), 1))
        assert float(full[:, V:].float().abs().max()) == 0.0, "padding columns must be zero (YTVLN_GEMM_A_ZERO_PADDED contract)"
    # the gradient buffer as the kernel leaves it (what LinearBf16Fn.backward receives)
    dl, ldd = ops._alloc_rows_bf16(M, V, dev)
    row_lse = torch.empty(M, device=dev); row_loss = torch.empty(M, device=dev); out = torch.empty(2, device=dev)
    one = torch.ones(1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("ytvln_ce_fwd_bf16", buf.data_ptr(), LD, tgt.data_ptr(), -1, row_lse.data_ptr(), row_loss.data_ptr(), out.data_ptr(), M, V, st)
    _lib.call("ytvln_ce_bwd_bf16", buf.data_ptr(), LD, tgt.data_ptr(), -1, row_lse.data_ptr(), out.data_ptr(), one.data_ptr(), dl.data_ptr(), ldd, M, V, st)
    full = torch.as_strided(dl, (M, ldd), (ldd, 1))
    assert ldd % 8 == 0 and ldd > V and torch.equal(full[:, :V], gb) and float(full[:, V:].float().abs().max()) == 0.0
    pred_b = rnd(dev, M, V, seed=3).to(BF).requires_grad_(True)
    pred = pred_b.detach().float().requires_grad_(True)
    t = torch.softmax(rnd(dev, M, V, seed=4), -1)
    mk = (torch.arange(M, device=dev) % 3 == 0).long()
    kb = ops.kl_masked(pred_b, t, mk)
    kb.backward()
    k32 = ops.kl_masked(pred, t, mk)
    k32.backward()
    assert float(kb) == float(k32) and pred_b.grad.dtype == BF and torch.equal(pred_b.grad, pred.grad.to(BF))
    # AdamW with the bf16 copy
    n = 5000
    p0, g, m0, v0 = rnd(dev, n, seed=5), rnd(dev, n, seed=6), 0.1 * rnd(dev, n, seed=7), (0.1 * rnd(dev, n, seed=8)).abs()
    chunks = torch.tensor([0, 0, 0], dtype=torch.int64)
    import struct
    table = torch.frombuffer(bytearray(struct.pack("<qqff", 0, 2048, 0.01, 0.0) + struct.pack("<qqff", 2048, n - 2048, 0.0, 0.0)), dtype=torch.uint8).to(dev)
    hyper = torch.tensor([0.9, 0.999, 1e-6, 1e-3, 1e-3, 0, 0, 0], device=dev)
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    pbf = torch.zeros(n, device=dev, dtype=BF)
    ops.adamw_step(pa, g, ma, va, table, 2, hyper)
    ops.adamw_step(pb, g, mb, vb, table, 2, hyper, p_bf16=pbf)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(pbf, pb.to(BF))


@pytest.mark.parametrize("precision,N,T,heads,d", [("bf16", 96, 80, 12, 64), ("bf16", 56, 288, 8, 128), ("bf16", 24, 576, 8, 128),
                                                    ("fp32", 56, 80, 12, 64), ("fp32", 56, 288, 8, 128)])
def test_attention_full_grids_every_block_right_and_reproducible(dev, lib, precision, N, T, heads, d):
    """The attention kernels at the grid sizes of BASELINE configs[1] / [3] / [4] (thousands of one-wave workgroups, several per CU), judged
    PER 32-query block: a relative-L2 over the whole tensor (test_attention_bf16) does not see one wrong block in 3500.  That is what a
    write-after-read race on a tile buffer produced in the bf16 forward kernel (the LDS-DMA of tile t+1 overtaking queued reads of tile t: 1-3
    blocks per launch off by 0.1-0.4, log-sum-exp right, different blocks every run) until b_reads_done() / lds_reads_done() made the kernels
    wait for their reads.  Checked here: every block of the forward output against an fp32 softmax on the same inputs (into buffers pre-filled
    with junk), and forward + backward bit-identical over repeated launches while a second HIP stream keeps the chip busy."""
    from ytvln import ops
    bf = precision == "bf16"
    g = torch.Generator().manual_seed(N + T + d)
    H = heads * d
    qkv = (torch.randn((N * T, 3 * H), generator=g) * 0.5).to(dev)
    qkv = (qkv.to(BF) if bf else qkv).requires_grad_()
    do = (torch.randn((N * T, H), generator=g) * 0.5).to(dev)
    do = do.to(BF) if bf else do
    mask = torch.zeros(N, T, device=dev)
    mask[:, T - 3:] = -10000.0
    q, k, v = [qkv.detach()[:, j * H:(j + 1) * H].float().view(N, T, heads, d).transpose(1, 2) for j in range(3)]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d) + mask[:, None, None, :]
    ref = torch.softmax(s, -1) @ v                                           # [N, heads, T, d]
    ref_lse = torch.logsumexp(s, -1)
    bar = 3e-2 if bf else 2e-5       # bf16: P and the output are rounded to bf16 (values ~0.3: errors up to ~5e-3 seen); a stale tile gives >= 0.1
    side = torch.cuda.Stream()
    A, B = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    runs = []
    for i in range(4):
        junk = torch.full((N * T, H), 7.0, device=dev, dtype=qkv.dtype)
        del junk                                                             # the kernel's output buffer starts as 7.0, not as the last result
        torch.cuda.synchronize()
        if i >= 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    ops.linear(A, B, None)
        qkv.grad = None
        out, lse = ops.SelfAttentionFn.apply(qkv, mask, N, T, heads, 0.0, None, 0)
        out.backward(do)
        torch.cuda.synchronize()
        o = out.detach().float().view(N, T, heads, d).transpose(1, 2)
        err = (o - ref).abs().amax(-1)                                       # [N, heads, T]
        worst = float(err.max())
        assert worst < bar, (i, worst, (err > bar).nonzero()[:8].tolist())
        assert float((lse - ref_lse).abs().max()) < 1e-2 if bf else 1e-4
        runs.append((out.detach().clone(), lse.detach().clone(), qkv.grad.clone()))
    for i in range(1, 4):
        for a, b, what in zip(runs[i], runs[0], ("out", "lse", "dqkv")):
            assert torch.equal(a, b), (i, what, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("H", [768, 1024, 36])
def test_embedding_dropout_masks_match_between_forward_and_backward_bf16(dev, lib, H):
    """on the bf16-resident path the embedding forwards (4-wide row kernels) and their LayerNorm backward (the 16-bytes-
    per-lane kernel whenever H % 8 == 0) must regenerate the SAME post-LayerNorm dropout mask.  The draw scheme is now a function of the row
    type and H alone (csrc/norm.hip drop4).  Check: with p > 0 the gradient that reaches the LayerNorm is zero exactly where the forward
    output is zero, and equals dy / (1 - p) elsewhere -- read off beta's gradient with one-hot dy, and off the full backward against an fp64
    LayerNorm backward fed with the mask taken from the forward output."""
    from ytvln import ops
    ops.set_matmul_precision("bf16")
    try:
        p, N, T, V = 0.25, 6, 20, 50
        drop = ops.DropoutState(dev)
        g = torch.Generator().manual_seed(H)
        ids = torch.randint(1, V, (N, T), generator=g).to(dev)
        word = torch.randn(V, H, generator=g).to(dev).requires_grad_(True)
        pos = torch.randn(T + 3, H, generator=g).to(dev).requires_grad_(True)
        typ = torch.randn(2, H, generator=g).to(dev).requires_grad_(True)
        gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
        beta = (0.5 + 0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)      # beta ~ 0.5: a kept output is never exactly 0
        y = ops.text_embed(ids, None, word, pos, typ, gamma, beta, 1e-12, p, drop)
        assert y.dtype == BF
        keep = (y.float() != 0)
        rate = 1.0 - float(keep.float().mean())
        assert abs(rate - p) < 0.03, rate
        dy = torch.randn(N, T, H, generator=g).to(dev).to(BF)
        y.backward(dy)
        # d beta = sum over rows of (dy * keep / (1 - p)): exact statement of "the backward used the forward's mask"
        ref_db = (dy.double() * keep.double() / (1 - p)).view(-1, H).sum(0)
        assert rel_l2(beta.grad, ref_db) < 2e-3, rel_l2(beta.grad, ref_db)
        wrong_db = (dy.double() / 1.0).view(-1, H).sum(0) * (1 - rate) / (1 - p)          # what an unrelated mask would give in expectation
        assert rel_l2(beta.grad, wrong_db) > 0.1
        # image side
        R = 12
        img = torch.randn(N, R, H, generator=g).to(dev).to(BF).requires_grad_(True)
        loc = torch.rand(N, R, 12, generator=g).to(dev)
        loc[..., 11] = torch.randint(0, 4, (N, R), generator=g).to(dev).float()
        W5, b5 = torch.randn(H, 5, generator=g).to(dev).requires_grad_(True), torch.randn(H, generator=g).to(dev).requires_grad_(True)
        W4, b4 = torch.randn(H, 4, generator=g).to(dev).requires_grad_(True), torch.randn(H, generator=g).to(dev).requires_grad_(True)
        W2, b2 = torch.randn(H, 2, generator=g).to(dev).requires_grad_(True), torch.randn(H, generator=g).to(dev).requires_grad_(True)
        E = torch.randn(4, H, generator=g).to(dev).requires_grad_(True)
        beta2 = (0.5 + 0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
        yi = ops.image_embed(img, loc, W5, b5, W4, b4, W2, b2, E, gamma.detach().requires_grad_(True), beta2, 1e-12, p, drop)
        keep_i = (yi.float() != 0)
        dyi = torch.randn(N, R, H, generator=g).to(dev).to(BF)
        yi.backward(dyi)
        ref_db2 = (dyi.double() * keep_i.double() / (1 - p)).view(-1, H).sum(0)
        assert rel_l2(beta2.grad, ref_db2) < 2e-3, rel_l2(beta2.grad, ref_db2)
    finally:
        ops.set_matmul_precision("fp32")


def _tile_max_err(C, ref, t=256):
    M, N = C.shape
    pm, pn = (-M) % t, (-N) % t
    d = torch.nn.functional.pad((C.double() - ref).abs(), (0, pn, 0, pm))
    return d.view((M + pm) // t, t, (N + pn) // t, t).amax(dim=(1, 3))


@pytest.mark.parametrize("M,N,K,ta,tb", [(129024, 1024, 1024, 0, 1), (17920, 3072, 768, 0, 1), (17920, 768, 3072, 0, 0), (1024, 1024, 129024, 1, 0),
                                         (129024, 1024, 1024, 0, 0)])
def test_gemm_bf16_full_grids_every_tile_right_and_reproducible(dev, lib, M, N, K, ta, tb):
    """the bf16 GEMM's LDS-DMA ring at the grid sizes of BASELINE configs[4] (224 pairs x 576 regions = 129024 image rows,
    17920 text rows), judged PER 256x256 output tile against fp64 on the same bf16 inputs, bit-identical over repeated launches, alone and
    beside a busy second stream (tools/bf16_repro.py promoted to a test: a write-after-read race on a ring slot shows up as a few wrong
    tiles, different ones every run, invisible to a whole-matrix relative L2)."""
    from ytvln import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn((K, M) if ta else (M, K), generator=g) * 0.5).to(dev).to(BF)
    B = (torch.randn((N, K) if tb else (K, N), generator=g) * 0.5).to(dev).to(BF)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    scale = float(ref.abs().max())
    side = torch.cuda.Stream()
    X, W = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    first = None
    for i in range(12):
        C = torch.full((M, N), 7.0, device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        if i >= 6:
            with torch.cuda.stream(side):
                for _ in range(4):
                    ops.linear(X, W, None)
        _gemm(ops, A, ta, B, tb, C, M, N, K)
        torch.cuda.synchronize()
        if i in (0, 6):
            err = _tile_max_err(C, ref)
            assert float(err.max()) < 3e-6 * scale * max(1.0, math.sqrt(K / 1024)), (i, float(err.max()), (err > 1e-5 * scale).nonzero()[:8].tolist())
        if first is None:
            first = C.clone()
        else:
            assert torch.equal(C, first), (i, float((C - first).abs().max()))


@pytest.mark.parametrize("rows,H,bf", [(16128, 1024, False), (4480, 768, False), (129024, 1024, True), (17920, 768, True)])
def test_layernorm_full_grids_every_row_right_and_reproducible(dev, lib, rows, H, bf):
    """VERDICT r4 item 6 for the row kernels: residual + LayerNorm forward and backward at the cfg-2 (fp32) and cfg-5 (bf16) row counts, every
    ROW against fp64 (row-wise max error, not a whole-tensor norm), bit-identical over repeated launches beside a busy second stream."""
    from ytvln import ops
    g = torch.Generator().manual_seed(rows + H)
    cast = (lambda t: t.to(BF)) if bf else (lambda t: t)
    x, res = cast(torch.randn(rows, H, generator=g).to(dev)), cast(torch.randn(rows, H, generator=g).to(dev))
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev).requires_grad_(True)
    dy = cast(torch.randn(rows, H, generator=g).to(dev))
    s = x.double() + res.double()
    mu, var = s.mean(-1, keepdim=True), s.var(-1, unbiased=False, keepdim=True)
    xh = (s - mu) / torch.sqrt(var + 1e-12)
    yr = gamma.detach().double() * xh + beta.detach().double()
    gg = dy.double() * gamma.detach().double()
    dsr = (gg - gg.mean(-1, keepdim=True) - xh * (gg * xh).mean(-1, keepdim=True)) / torch.sqrt(var + 1e-12)
    side = torch.cuda.Stream()
    X, W = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    first = None
    for i in range(8):
        xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        gamma.grad = beta.grad = None
        torch.cuda.synchronize()
        if i >= 4:
            with torch.cuda.stream(side):
                for _ in range(4):
                    ops.linear(X, W, None)
        y = ops.add_layer_norm(xr, rr, gamma, beta, 1e-12)
        y.backward(dy)
        torch.cuda.synchronize()
        if i in (0, 4):
            ey = (y.detach().double() - yr).abs().amax(-1)
            ed = (xr.grad.double() - dsr).abs().amax(-1)
            by, bd = (4e-2, 8e-2) if bf else (2e-5, 1e-4)          # bf16: one rounding of values up to ~5 / gradients up to ~10
            assert float(ey.max()) < by and float(ed.max()) < bd, (i, float(ey.max()), float(ed.max()), (ey > by).nonzero()[:4].tolist())
        cur = (y.detach().clone(), xr.grad.clone(), gamma.grad.clone(), beta.grad.clone())
        if first is None:
            first = cur
        else:
            for a, b, what in zip(cur, first, ("y", "dx", "dgamma", "dbeta")):
                assert torch.equal(a, b), (i, what)


@pytest.mark.parametrize("Tq,Tk", [(80, 576), (576, 80), (80, 252)])
def test_bf16_near_uniform_attention_gradient_bias_is_the_rounded_context(dev, lib, Tq, Tk):
    """VERDICT r4 "weak": the 10 % bar on the gradient norms of BertBiAttention's query / key projections in bf16 mode.  A co-attention direction
    that attends almost uniformly over 252-576 regions has dS = P o (dP - delta) with |dP - delta| << |delta|, and delta = sum_d O o dO is computed
    in backward from the context O the forward STORED AS bf16: the 2^-9 relative rounding of O is an error of the size of the whole of dP - delta.
    Pinned here at the kernel: against fp64 on the same bf16 inputs AND the same bf16-rounded O (what the kernel is given) dQ / dK agree to bf16
    rounding noise (relative L2 < 3e-2, norms within 2 %); against fp64 with the exact O the same gradients are off by far more -- the bias is the
    storage format of the context, not the kernel's arithmetic."""
    from ytvln import ops
    N, heads, d = 4, 8, 128
    H = heads * d
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    q = (torch.randn(N * Tq, H, generator=g) * 0.15).to(dev).to(BF)          # small logits: near-uniform attention
    k = (torch.randn(N * Tk, H, generator=g) * 0.15).to(dev).to(BF)
    # values that differ little between the regions of a pair (a common row + 3 % of noise): then dP_ij = dO_i . V_j is almost the same for every j,
    # i.e. almost delta_i -- the regime of the model's co-attention layers
    v = (torch.randn(N, 1, H, generator=g) + 0.03 * torch.randn(N, Tk, H, generator=g)).reshape(N * Tk, H).to(dev).to(BF)
    dout = torch.randn(N * Tq, H, generator=g).to(dev).to(BF)
    mask = torch.zeros(N, Tk, device=dev)
    out = torch.empty(N * Tq, H, device=dev, dtype=BF)
    scale = 1 / math.sqrt(d)
    lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, heads, Tq, Tk, d, scale, 0.0, None, 0)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, heads, Tq, Tk, d, scale, 0.0, None, 0)
    # fp64 backward written out, with delta from (a) the exact context, (b) the bf16 context the kernel reads
    qh, kh, vh, gh = (t.double().view(N, -1, heads, d).permute(0, 2, 1, 3) for t in (q, k, v, dout))
    P = torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1)
    O = P @ vh
    dP = gh @ vh.transpose(-1, -2)

    def grads(Octx):
        delta = (Octx * gh).sum(-1, keepdim=True)
        dS = P * (dP - delta)
        return (dS @ kh) * scale, (dS.transpose(-1, -2) @ qh) * scale

    Ob = out.double().view(N, Tq, heads, d).permute(0, 2, 1, 3)
    dq_exact, dk_exact = grads(O)
    dq_rounded, dk_rounded = grads(Ob)
    got_q = dq.double().view(N, Tq, heads, d).permute(0, 2, 1, 3)
    got_k = dk.double().view(N, Tk, heads, d).permute(0, 2, 1, 3)
    for name, got, ref_r, ref_e in (("dq", got_q, dq_rounded, dq_exact), ("dk", got_k, dk_rounded, dk_exact)):
        e_r, e_e = rel_l2(got, ref_r), rel_l2(got, ref_e)
        n_r = abs(float(got.norm() / ref_r.norm()) - 1.0)
        print(name, 'vs rounded-context ref', e_r, 'norm off', n_r, '| vs exact-context ref', e_e, 'norm ratio', float(got.norm() / ref_e.norm()))
        assert e_r < 3e-2 and n_r < 2e-2, (name, e_r, n_r)
        assert e_e > 2.0 * e_r, (name, e_r, e_e)          # the exact-context reference is the one that is far away
