"""Persistent / stream-K fp32 GEMM (csrc/gemm_sk.hip, ytvln_gemm_f32_sk) against the fp64 product, tile by tile over full cfg-2 grids.

Every check is per 128x128 output tile (a wrong tile in a 4000-tile launch moves a whole-matrix relative L2 by 1e-4: the LDS write-after-
read race of round 4 lived under such tests for two rounds), the whole-tile form must equal the launch-per-tile kernel bit for bit (same k
order per tile), the stream-K form must be bit-reproducible over 30 launches (fixed-order fix-up) and both leave the control block zero.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(dev, *shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dev)


class _Opts:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        from ytvln import _lib
        self.prev = {k: _lib.set_option(k, v) for k, v in self.kw.items()}

    def __exit__(self, *a):
        from ytvln import _lib
        for k, v in self.prev.items():
            _lib.set_option(k, v)


def _tile_errors(C, ref, tm=128, tn=128):
    """max |C - ref| per tile, relative to the tile's own rms of ref."""
    M, N = C.shape
    pm, pn = (-M) % tm, (-N) % tn
    d = torch.nn.functional.pad((C.double() - ref).abs(), (0, pn, 0, pm))
    r = torch.nn.functional.pad(ref * ref, (0, pn, 0, pm))
    d = d.view((M + pm) // tm, tm, (N + pn) // tn, tn).amax(dim=(1, 3))
    r = r.view((M + pm) // tm, tm, (N + pn) // tn, tn).mean(dim=(1, 3)).sqrt()
    return d / r.clamp_min(1e-30)


def _ctl_zero(dev):
    from ytvln import ops
    torch.cuda.synchronize()
    return all(int(t.abs().sum()) == 0 for t in ops._SK_CTL[dev][:2])


SHAPES = [   # M, N, K, transB, epilogue      (cfg-2 shapes: image rows 16128, text rows 4480)
    (16128, 1024, 1024, 1, 0), (16128, 1024, 1024, 0, 0), (16128, 3072, 1024, 1, 1), (16128, 1024, 3072, 0, 3),
    (4480, 3072, 768, 1, 1), (4480, 768, 3072, 1, 0), (4480, 768, 3072, 0, 0), (4480, 2304, 768, 1, 0), (4480, 768, 768, 1, 0),
    (16128, 1601, 1024, 1, 0), (2100, 1032, 160, 0, 2),
]


@pytest.mark.parametrize("M,N,K,tb,epi", SHAPES)
@pytest.mark.parametrize("form", ["dp", "sk"])
@pytest.mark.parametrize("tile", [4, 3, 0])
def test_persistent_gemm_every_tile_right(dev, lib, M, N, K, tb, epi, form, tile):
    from ytvln import ops
    if not tb and N % 4:
        pytest.skip("N-contiguous B needs N % 4 == 0 on the LDS-DMA path")
    A = _rand(dev, M, K, seed=M + K)
    B = _rand(dev, *((N, K) if tb else (K, N)), seed=N + 7 * K)
    bias = None if epi == 3 else _rand(dev, N, seed=3)
    aux_in = _rand(dev, M, N, seed=11) if epi == 3 else None
    ref = A.double() @ (B.double().t() if tb else B.double())
    if bias is not None:
        ref = ref + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref.clamp(min=0)
    elif epi == 3:
        z = aux_in.double()
        ref = ref * (0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * math.pi) ** 0.5)

    def run(**opts):
        C = torch.full((M, N), float("nan"), device=dev)
        aux = aux_in.clone() if epi == 3 else (torch.empty(M, N, device=dev) if epi == 1 else None)
        with _Opts(**opts):
            ops._gemm(A, K, 0, B, B.stride(0), tb, C, N, M, N, K, bias=bias, aux=aux, ldaux=N, epi=epi)
        torch.cuda.synchronize()
        return C

    import ctypes
    vals = [ctypes.c_int(0) for _ in range(5)]
    with _Opts(GEMM_SK={"dp": 2, "sk": 3}[form], GEMM_SK_TILE=tile):
        lib.ytvln_gemm_sk_plan(M, N, K, 0, epi, *[ctypes.byref(v) for v in vals])
    if not vals[0].value:
        pytest.skip("shape not eligible for this form of the persistent kernel")
    C = run(GEMM_SK={"dp": 2, "sk": 3}[form], GEMM_SK_TILE=tile)
    assert torch.isfinite(C).all(), "a tile was never written"
    err = _tile_errors(C, ref)
    bar = 4e-6 * math.sqrt(K) + 1e-6
    assert float(err.max()) < bar, (float(err.max()), bar, torch.nonzero(err >= bar)[:8].tolist())
    assert _ctl_zero(dev), "control block not left zero"
    if form == "dp":        # same k order per tile as the launch-per-tile kernel on the same tile shape: bit-identical
        C0 = run(GEMM_SK=0, GEMM_TILE=tile, GEMM_SPLITS=1)
        if tile == 0 and epi == 3:
            # round 6: the 128x128 persistent kernel (the one GEMM kernel with a scratch segment) takes the compiler-counted epilogue loads for
            # epilogues that read a matrix (gemm_epilogue<..., HAND = false>): same accumulators, the x GELU' product contracted differently
            assert float((C - C0).abs().max()) <= 1e-6 * float(C0.abs().max()) + 1e-4, float((C - C0).abs().max())
        else:
            assert torch.equal(C, C0), float((C - C0).abs().max())
    else:                   # stream-K: fixed-order fix-up -> the same bits every time
        for _ in range(10):
            assert torch.equal(C, run(GEMM_SK=3, GEMM_SK_TILE=tile))
        assert _ctl_zero(dev)


def test_persistent_gemm_zero_padded_k_tail(dev, lib):
    """dX of the 30522-wide decoder (vilbert.py:906 backward): K = 30522 with A's K tail zero padded, B's k rows clamped."""
    from ytvln import ops
    from ytvln._lib import GEMM_A_ZERO_PADDED
    M, V, H = 4480, 30522, 768
    ld = (V + 31) // 32 * 32
    dl = torch.zeros(M, ld, device=dev)
    dl[:, :V] = _rand(dev, M, V, seed=1) * 0.1
    E = _rand(dev, V, H, seed=2)
    ref = dl[:, :V].double() @ E.double()
    for form in (2, 3):
        dx = torch.empty(M, H, device=dev)
        with _Opts(GEMM_SK=form):
            ops._gemm(dl, ld, 0, E, H, 0, dx, H, M, H, V, flags=GEMM_A_ZERO_PADDED)
        assert float(_tile_errors(dx, ref).max()) < 4e-6 * math.sqrt(V) + 1e-6, form
    assert _ctl_zero(dev)


@pytest.mark.timeout(120)
def test_persistent_gemm_two_streams_at_once(dev, lib):
    """Two stream-K launches interleaving on the CUs (the text and the image side of TwoStream): tickets make every wait point at a
    workgroup that has already started, so this must neither hang nor change a bit."""
    from ytvln import ops
    M, N, K = 4480, 3072, 768
    A, W = _rand(dev, M, K, seed=1), _rand(dev, N, K, seed=2)
    A2, W2 = _rand(dev, 16128, 1024, seed=3), _rand(dev, 1024, 1024, seed=4)
    with _Opts(GEMM_SK=3):
        C_ref = torch.empty(M, N, device=dev)
        ops._gemm(A, K, 0, W, K, 1, C_ref, N, M, N, K)
        D_ref = torch.empty(16128, 1024, device=dev)
        ops._gemm(A2, 1024, 0, W2, 1024, 1, D_ref, 1024, 16128, 1024, 1024)
        torch.cuda.synchronize()
        side = ops.TwoStream.side_stream(dev)
        Cs = [torch.empty(M, N, device=dev) for _ in range(20)]
        Ds = [torch.empty(16128, 1024, device=dev) for _ in range(20)]
        side.wait_stream(torch.cuda.current_stream())
        for i in range(20):
            ops._gemm(A2, 1024, 0, W2, 1024, 1, Ds[i], 1024, 16128, 1024, 1024)
            with torch.cuda.stream(side):
                ops._gemm(A, K, 0, W, K, 1, Cs[i], N, M, N, K)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    assert all(torch.equal(c, C_ref) for c in Cs) and all(torch.equal(d, D_ref) for d in Ds)
    assert _ctl_zero(dev)


@pytest.mark.parametrize("M,N,K,tb,epi", [(4480, 3072, 768, 1, 1), (4480, 3072, 768, 0, 3), (4480, 30528, 768, 1, 0), (4480, 768, 3072, 1, 0),
                                          (4480, 2304, 768, 1, 0), (4500, 1032, 160, 0, 2), (16128, 1024, 1024, 1, 0), (300, 520, 96, 1, 0)])
@pytest.mark.parametrize("tile,bm", [(5, 224), (6, 160)])
def test_gemm_224_row_tile_every_tile_right(dev, lib, M, N, K, tb, epi, tile, bm):
    """The 224x256 / 160x256 tiles of gemm_dma_kernel (eight waves of 224x32 / 160x32; 28 / 20 LDS-DMA pieces of A over 8 waves: the last waves own fewer): every
    128x128 output tile against fp64, ragged M / N (boundary epilogue, clamped DMA rows), fused epilogues, bit-reproducible, and -- the k order
    per accumulator being the same -- bit-identical to the 256x256 tile."""
    from ytvln import ops
    A = _rand(dev, M, K, seed=M + K)
    B = _rand(dev, *((N, K) if tb else (K, N)), seed=N + 7 * K)
    bias = None if epi == 3 else _rand(dev, N, seed=3)
    aux_in = _rand(dev, M, N, seed=11) if epi == 3 else None
    ref = A.double() @ (B.double().t() if tb else B.double())
    if bias is not None:
        ref = ref + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref.clamp(min=0)
    elif epi == 3:
        z = aux_in.double()
        ref = ref * (0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * math.pi) ** 0.5)

    def run(**opts):
        C = torch.full((M, N), float("nan"), device=dev)
        aux = aux_in.clone() if epi == 3 else (torch.empty(M, N, device=dev) if epi == 1 else None)
        with _Opts(**opts):
            ops._gemm(A, K, 0, B, B.stride(0), tb, C, N, M, N, K, bias=bias, aux=aux, ldaux=N, epi=epi)
        torch.cuda.synchronize()
        return C

    import ctypes
    tm, tn, sp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    with _Opts(GEMM_T224=1, GEMM_TILE=tile, GEMM_SPLITS=1, GEMM_SK=0):
        lib.ytvln_gemm_plan(M, N, K, 0, epi, ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sp))
    assert (tm.value, tn.value, sp.value) == (bm, 256, 1)
    C = run(GEMM_T224=1, GEMM_TILE=tile, GEMM_SPLITS=1, GEMM_SK=0)
    assert torch.isfinite(C).all(), "a tile was never written"
    err = _tile_errors(C, ref)
    bar = 4e-6 * math.sqrt(K) + 1e-6
    assert float(err.max()) < bar, (float(err.max()), bar, torch.nonzero(err >= bar)[:8].tolist())
    for _ in range(5):
        assert torch.equal(C, run(GEMM_T224=1, GEMM_TILE=tile, GEMM_SPLITS=1, GEMM_SK=0))
    C4 = run(GEMM_TILE=4, GEMM_SPLITS=1, GEMM_SK=0)
    assert torch.equal(C, C4), float((C - C4).abs().max())


@pytest.mark.parametrize("M,N,K,tb", [(4480, 768, 3072, 1), (4480, 768, 3072, 0), (8960, 768, 3072, 1), (4480, 768, 2304, 0)])
def test_gemm_small_row_tiles_split_k(dev, lib, M, N, K, tb):
    """Split-K on the 160- / 224-row tiles (the planner's choice for the N = 768 text shapes: 84 tiles x 3 splits in one round): every 128x128
    output tile against fp64, bias applied once by the fixed-order reduce, the same bits every launch."""
    from ytvln import ops
    import ctypes
    A = _rand(dev, M, K, seed=M + K)
    B = _rand(dev, *((N, K) if tb else (K, N)), seed=N + 7 * K)
    bias = _rand(dev, N, seed=3)
    ref = A.double() @ (B.double().t() if tb else B.double()) + bias.double()
    tm, tn, sp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    lib.ytvln_gemm_plan(M, N, K, 0, 0, ctypes.byref(tm), ctypes.byref(tn), ctypes.byref(sp))
    forced = {} if (tm.value in (160, 224) and sp.value > 1) else dict(GEMM_TILE=6, GEMM_SPLITS=3)

    def run():
        C = torch.full((M, N), float("nan"), device=dev)
        with _Opts(GEMM_SK=0, **forced):
            ops._gemm(A, K, 0, B, B.stride(0), tb, C, N, M, N, K, bias=bias)
        torch.cuda.synchronize()
        return C

    C = run()
    assert torch.isfinite(C).all()
    err = _tile_errors(C, ref)
    bar = 4e-6 * math.sqrt(K) + 1e-6
    assert float(err.max()) < bar, (float(err.max()), bar, torch.nonzero(err >= bar)[:8].tolist())
    for _ in range(5):
        assert torch.equal(C, run())
