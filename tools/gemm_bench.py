import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
shapes = [  # M, N, K, tA, tB
    (16128, 1024, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (1024, 1024, 16128, 1, 0), (16128, 3072, 1024, 0, 1),
    (4480, 768, 3072, 0, 1), (4480, 3072, 768, 0, 1), (4480, 768, 768, 0, 1), (768, 3072, 4480, 1, 0), (4480, 768, 3072, 0, 0)]
def run(M, N, K, ta, tb, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    lda = M if ta else K; ldb = K if tb else N
    for _ in range(3): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9
for v in os.environ.get("VARIANTS", "0").split(","):
    os.environ["YTVLN_GEMM_UNUSED"] = v
    print("variant", v, " ".join(f"{run(*s):6.1f}" for s in shapes), flush=True)
