"""Per-shape GEMM rates at the cfg-2 shapes (HIP events, 20 launches each).  Planner knobs are process-wide environment variables
(YTVLN_GEMM_TILE, YTVLN_GEMM_SPLITS: the options of include/ytvln.h, read from the environment at first use): run once per setting.  SHAPES=wgrad|dx|fwd|all."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
if os.environ.get("PRECISION"):
    ops.set_matmul_precision(os.environ["PRECISION"])      # fp32x3 | bf16
SETS = {
    "wgrad": [(1024, 1024, 16128, 1, 0), (3072, 1024, 16128, 1, 0), (2048, 1024, 16128, 1, 0), (768, 3072, 4480, 1, 0), (3072, 768, 4480, 1, 0),
              (2304, 768, 4480, 1, 0), (768, 768, 4480, 1, 0), (30522, 768, 4480, 1, 0), (30528, 768, 4480, 1, 0), (1024, 2048, 16128, 1, 0)],
    "dx": [(16128, 1024, 1024, 0, 0), (16128, 1024, 3072, 0, 0), (4480, 768, 3072, 0, 0), (4480, 3072, 768, 0, 0), (4480, 768, 2304, 0, 0),
           (4480, 768, 768, 0, 0)],
    "fwd": [(16128, 1024, 1024, 0, 1), (16128, 3072, 1024, 0, 1), (4480, 768, 3072, 0, 1), (4480, 3072, 768, 0, 1)],
}
SETS["fwddx"] = [s for s in SETS["dx"] + SETS["fwd"] if s[0] == 4480] + [(4480, 2304, 768, 0, 1), (4480, 1024, 768, 0, 1), (4480, 768, 1024, 0, 1), (4480, 768, 1024, 0, 0)]
SETS["cfg4"] = [(24192, 1024, 1024, 0, 1), (24192, 1024, 1024, 0, 0), (24192, 3072, 1024, 0, 1), (24192, 1024, 3072, 0, 0), (24192, 2048, 1024, 0, 1),
                (7680, 768, 768, 0, 1), (7680, 2304, 768, 0, 1), (7680, 3072, 768, 0, 1), (7680, 768, 3072, 0, 1), (7680, 768, 3072, 0, 0), (7680, 1024, 768, 0, 1)]
SETS["probe"] = [(16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (4480, 768, 3072, 0, 1), (4480, 3072, 768, 0, 1), (4480, 2304, 768, 0, 1), (1024, 1024, 16128, 1, 0), (768, 768, 4480, 1, 0)]
SETS["probe2"] = [(16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (16128, 3072, 1024, 0, 1), (4480, 768, 3072, 0, 1), (4480, 768, 3072, 0, 0),
                  (4480, 3072, 768, 0, 1), (4480, 2304, 768, 0, 1), (4480, 768, 768, 0, 1), (1024, 1024, 16128, 1, 0), (768, 3072, 4480, 1, 0), (768, 768, 4480, 1, 0),
                  (2048, 2048, 4096, 0, 1), (4096, 4096, 4096, 0, 1)]
SETS["probe3"] = [(4096, 4096, 4096, 0, 1), (4096, 4096, 4096, 0, 0), (4096, 4096, 4096, 1, 0), (2048, 2048, 8192, 0, 1), (16128, 1024, 1024, 0, 1), (4480, 3072, 768, 0, 1)]
SETS["probe4"] = [(4096, 4096, 4096, 0, 1), (4096, 4096, 4096, 1, 0)]
SETS["probe5"] = [(4096, 4096, 4096, 0, 1)]
SETS["probe6"] = [(16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (16128, 3072, 1024, 0, 1), (16128, 1024, 3072, 0, 0), (1024, 1024, 16128, 1, 0), (3072, 1024, 16128, 1, 0), (4096, 4096, 4096, 0, 1), (4480, 3072, 768, 0, 1)]
SETS["img"] = [(16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (16128, 3072, 1024, 0, 1), (1024, 1024, 16128, 1, 0)]
SETS["img1"] = [(16128, 1024, 1024, 0, 1)]
SETS["big1"] = [(4096, 4096, 4096, 0, 1)]
which = os.environ.get("SHAPES", "all")
shapes = sum(SETS.values(), []) if which == "all" else SETS[which]


MODE = os.environ.get("MODE", "plain")        # plain | bias | gelu (bias + erf-GELU + pre-activation store) | beta (C += ...) | dgelu (x GELU'(aux))
COLD = int(os.environ.get("COLD", "0"))        # > 0: rotate over that many operand / output sets (defeats the 256 MB memory-side cache)
from ytvln._lib import EPI_GELU, EPI_MUL_DGELU


def run(M, N, K, ta, tb, iters=int(os.environ.get("ITERS", "20"))):
    nset = max(1, COLD)
    fill = {"randn": torch.randn, "zeros": torch.zeros, "ones": torch.ones}[os.environ.get("DATA", "randn")]      # operand data: the chip's clock under load depends on it
    As = [fill((K, M) if ta else (M, K), device=dev) for _ in range(nset)]
    B = fill((N, K) if tb else (K, N), device=dev)
    Cs = [torch.randn(M, N, device=dev) for _ in range(nset)]
    aux = [torch.randn(M, N, device=dev) for _ in range(nset)] if MODE in ("gelu", "dgelu") else [None] * nset
    bias = torch.randn(N, device=dev) if MODE in ("bias", "gelu") else None
    lda = M if ta else K; ldb = K if tb else N
    kw = {"plain": {}, "bias": dict(bias=bias), "gelu": dict(bias=bias, epi=EPI_GELU, ldaux=N), "beta": dict(beta=1.0),
          "dgelu": dict(epi=EPI_MUL_DGELU, ldaux=N)}[MODE]

    def go(i):
        j = i % nset
        extra = dict(kw)
        if MODE in ("gelu", "dgelu"): extra["aux"] = aux[j]
        ops._gemm(As[j], lda, ta, B, ldb, tb, Cs[j], N, M, N, K, **extra)
    for i in range(3): go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): go(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1000, 2.0 * M * N * K / ms / 1e9


for s in shapes:
    us, tf = run(*s)
    print(f"{s[0]:6d} {s[1]:6d} {s[2]:6d} tA{s[3]} tB{s[4]}  {us:8.1f} us  {tf:6.1f} TF/s", flush=True)
