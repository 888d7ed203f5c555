"""bf16 weight-gradient launches (A = dY^T and B = X both k-major, fp32 C) under forced plans: GEMM_TILE 0 (128x128) / 4 (256x256) x GEMM_SPLITS,
against the planner's own choice ("plan"), interleaved passes with warm clocks.  SHAPES="M,N,K;..." overrides the cfg-5 list."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
T, R = 224 * 80, 224 * 576
shapes = [(768, 768, T), (1024, 768, T), (768, 1024, T), (2048, 768, T), (2304, 768, T), (768, 3072, T), (3072, 768, T), (1024, 1024, R), (2048, 1024, R), (3072, 1024, R)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(x) for x in t.split(",")) for t in os.environ["SHAPES"].split(";")]
heat = torch.randn(8192, 8192, device=dev).bfloat16()
def warm():
    for _ in range(12): torch.matmul(heat, heat)
def once(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for (M, N, K) in shapes:
    dY = torch.randn(K, M, device=dev).bfloat16(); X = torch.randn(K, N, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev)
    go = lambda: ops._gemm_bf16(dY, M, 1, X, N, 0, C, N, M, N, K)
    cfgs = [("plan", -1, -1)] + [(f"t{t}s{sp}", t, sp) for t in (0, 4) for sp in (2, 4, 7, 8, 14, 16, 28) if K // sp >= 512]
    best = {c[0]: 1e30 for c in cfgs}
    for p in range(2):
        warm()
        for name, t, sp in (cfgs if p % 2 == 0 else cfgs[::-1]):
            _lib.set_option("GEMM_TILE", t); _lib.set_option("GEMM_SPLITS", sp)
            ops._BF16_WS_CACHE.clear()
            try:
                go(); best[name] = min(best[name], once(go))
            finally:
                _lib.set_option("GEMM_TILE", -1); _lib.set_option("GEMM_SPLITS", -1)
    fl = 2.0 * M * N * K
    order = sorted(best, key=lambda k: best[k])
    print(f"{M:5d} {N:5d} {K:7d}  plan {best['plan']:7.1f}us {fl / best['plan'] / 1e6:6.0f} TF | best " + "  ".join(f"{k} {best[k]:.1f}" for k in order[:5]), flush=True)
