"""What the fused epilogues of the fp32 GEMM cost at the cfg-2 FFN shapes (see tools/gemm_bf16_epilogues.py for the bf16 path)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
heat = torch.randn(8192, 8192, device=dev)
def warm():
    for _ in range(6): torch.matmul(heat, heat)
def once(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for (M, N, K) in [(16128, 1024, 1024), (4480, 3072, 768)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev); Z = torch.empty(M, N, device=dev)
    dY = torch.randn(M, N, device=dev); Wt = torch.randn(N, K, device=dev); dX = torch.empty(M, K, device=dev); Zk = torch.randn(M, K, device=dev)
    cases = {
        "plain": lambda: ops._gemm(A, K, 0, W, K, 1, C, N, M, N, K),
        "bias": lambda: ops._gemm(A, K, 0, W, K, 1, C, N, M, N, K, bias=b),
        "bias+gelu+z": lambda: ops._gemm(A, K, 0, W, K, 1, C, N, M, N, K, bias=b, aux=Z, ldaux=N, epi=ops.EPI_GELU),
        "bias+relu": lambda: ops._gemm(A, K, 0, W, K, 1, C, N, M, N, K, bias=b, epi=ops.EPI_RELU),
        "dX plain": lambda: ops._gemm(dY, N, 0, Wt, K, 0, dX, K, M, K, N),
        "dX x gelu'(z)": lambda: ops._gemm(dY, N, 0, Wt, K, 0, dX, K, M, K, N, aux=Zk, ldaux=K, epi=ops.EPI_MUL_DGELU),
    }
    best = {k: 1e30 for k in cases}
    names = list(cases)
    for p in range(3):
        warm()
        for k in (names if p % 2 == 0 else names[::-1]):
            cases[k](); best[k] = min(best[k], once(cases[k]))
    print(f"{M}x{N}x{K}: " + "  ".join(f"{k} {best[k]:.1f}us" for k in names), flush=True)
