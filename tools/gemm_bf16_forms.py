"""bf16 GEMM main-loop forms (run-time option GEMM_BF16_FORM: where the next tiles' LDS-DMA is issued) and the first-round stagger
(GEMM_STAGGER) on the BASELINE configs[4] shapes, against each other and against the vendor library (torch.matmul), in ONE process under the
fair protocol of tools/gemm_sk_bench.py: every pass starts behind a burst of GEMM work, the configurations are interleaved, the order
reverses every pass, minimum over the passes.  CONFIGS="form:stagger,..." (default 0:0,1:0,2:0,0:50,2:50); SHAPES filters by substring;
KSWEEP=1 adds the K sweep at the image shape (fixed cost per round and per-k-tile slope of every configuration)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
R, T = 224 * 576, 224 * 80
shapes = [("img fwd 1024->1024", R, 1024, 1024, 0, 1), ("img fwd 1024->3072", R, 3072, 1024, 0, 1), ("img fwd 2048->1024", R, 1024, 2048, 0, 1),
          ("img dX 1024<-1024", R, 1024, 1024, 0, 0), ("img dX 1024<-3072", R, 1024, 3072, 0, 0),
          ("img dW 1024x1024", 1024, 1024, R, 1, 0), ("img dW 3072x1024", 3072, 1024, R, 1, 0),
          ("txt fwd 768->2304", T, 2304, 768, 0, 1), ("txt fwd 768->3072", T, 3072, 768, 0, 1), ("txt fwd 3072->768", T, 768, 3072, 0, 1),
          ("txt fwd 768->768", T, 768, 768, 0, 1), ("txt dX 768<-3072", T, 768, 3072, 0, 0), ("txt dX 3072<-768", T, 3072, 768, 0, 0),
          ("txt dW 3072x768", 3072, 768, T, 1, 0), ("lm decoder 768->30522", T, 30522, 768, 0, 1)]
only = os.environ.get("SHAPES")
configs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "0:0,1:0,2:0,0:50,2:50").split(",")]
heat = torch.randn(8192, 8192, device=dev).bfloat16()


def warm():
    for _ in range(12):
        torch.matmul(heat, heat)


def once(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


def setcfg(c):
    _lib.set_option("GEMM_BF16_FORM", c[0])
    _lib.set_option("GEMM_STAGGER", c[1])


def race(fs, n=8, passes=3):
    """fs: list of callables (setup + launch); returns min us per callable over interleaved passes"""
    best = [1e30] * len(fs)
    order = list(range(len(fs)))
    for p in range(passes):
        warm()
        for i in (order if p % 2 == 0 else order[::-1]):
            fs[i](1)
            best[i] = min(best[i], fs[i](n))
    return best


hdr = " ".join(f"f{c[0]}s{c[1]:<3d}" for c in configs)
print(f"{'shape':24s} {'M':>7s} {'N':>6s} {'K':>7s}   {hdr}   torch   best/torch (us; TF/s of the best form)", flush=True)
for name, M, N, K, ta, tb in shapes:
    if only and not any(o in name for o in only.split(",")):
        continue
    A = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
    cdt = torch.float32 if ta else torch.bfloat16
    ldc = (N + 7) // 8 * 8
    C = torch.empty(M, ldc, device=dev, dtype=cdt)
    Ct = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)

    def mk(c):
        def f(n):
            setcfg(c)
            return once(lambda: ops._gemm_bf16(A, A.stride(0), ta, B, B.stride(0), tb, C, ldc, M, N, K), n)
        return f
    fs = [mk(c) for c in configs] + [lambda n: once(lambda: torch.matmul(At, Bt, out=Ct), n)]
    t = race(fs)
    # correctness of every form against the library result (bf16 rounding of the output only)
    ref = Ct.float()
    errs = []
    for c in configs:
        setcfg(c)
        C.zero_()
        ops._gemm_bf16(A, A.stride(0), ta, B, B.stride(0), tb, C, ldc, M, N, K)
        errs.append(float((C[:, :N].float() - ref).abs().max() / ref.abs().max()))
    best = min(t[:-1])
    fl = 2.0 * M * N * K
    print(f"{name:24s} {M:7d} {N:6d} {K:7d}   " + " ".join(f"{u:7.1f}" for u in t[:-1]) + f"  {t[-1]:7.1f}   {t[-1] / best:4.2f}x ({fl / best / 1e6:5.0f} TF/s)"
          f"   max rel diff {max(errs):.1e}", flush=True)
setcfg((0, 0))

if os.environ.get("KSWEEP"):
    M, N = R, 1024
    Ks = [256, 512, 1024, 2048]
    rounds = -(-M // 256) * -(-N // 256) / 256.0
    res = {c: [] for c in configs}
    lib = []
    for K in Ks:
        A = torch.randn(M, K, device=dev).bfloat16()
        B = torch.randn(N, K, device=dev).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        Ct = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def mk(c):
            def f(n):
                setcfg(c)
                return once(lambda: ops._gemm_bf16(A, K, 0, B, K, 1, C, N, M, N, K), n)
            return f
        t = race([mk(c) for c in configs] + [lambda n: once(lambda: torch.matmul(A, B.t(), out=Ct), n)])
        for c, u in zip(configs, t):
            res[c].append(u)
        lib.append(t[-1])
    for c in configs + ["lib"]:
        ys = lib if c == "lib" else res[c]
        pk, fx = np.polyfit(Ks, ys, 1)
        print(f"ksweep {str(c):10s} " + " ".join(f"{k}:{u:6.1f}" for k, u in zip(Ks, ys)) + f" | fixed {fx / rounds:5.2f} us/round, {pk * 64 / rounds:5.3f} us per 64-deep k-tile", flush=True)
    setcfg((0, 0))
