"""What the fused epilogues of the bf16 GEMM cost at the cfg-5 text FFN shapes: plain / bias / bias+GELU with the saved pre-activation (forward
FFN-in), and the GELU' input-gradient epilogue -- each against the plain product of the same shape (interleaved passes, warm clocks)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
heat = torch.randn(8192, 8192, device=dev).bfloat16()
def warm():
    for _ in range(12): torch.matmul(heat, heat)
def once(f, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for (M, N, K) in [(17920, 3072, 768), (129024, 1024, 1024), (17920, 768, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Z = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dY = torch.randn(M, N, device=dev).bfloat16(); Wt = torch.randn(N, K, device=dev).bfloat16(); dX = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    Zk = torch.randn(M, K, device=dev).bfloat16()
    cases = {
        "plain": lambda: ops._gemm_bf16(A, K, 0, W, K, 1, C, N, M, N, K),
        "bias": lambda: ops._gemm_bf16(A, K, 0, W, K, 1, C, N, M, N, K, bias=b),
        "bias+gelu+z": lambda: ops._gemm_bf16(A, K, 0, W, K, 1, C, N, M, N, K, bias=b, aux=Z, ldaux=N, epi=ops.EPI_GELU),
        "bias+relu": lambda: ops._gemm_bf16(A, K, 0, W, K, 1, C, N, M, N, K, bias=b, epi=ops.EPI_RELU),
        "dX plain": lambda: ops._gemm_bf16(dY, N, 0, Wt, K, 0, dX, K, M, K, N),
        "dX x gelu'(z)": lambda: ops._gemm_bf16(dY, N, 0, Wt, K, 0, dX, K, M, K, N, aux=Zk, ldaux=K, epi=ops.EPI_MUL_DGELU),
    }
    best = {k: 1e30 for k in cases}
    names = list(cases)
    for p in range(3):
        warm()
        for k in (names if p % 2 == 0 else names[::-1]):
            cases[k](); best[k] = min(best[k], once(cases[k]))
    print(f"{M}x{N}x{K}: " + "  ".join(f"{k} {best[k]:.1f}us" for k in names), flush=True)
