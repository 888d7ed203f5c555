"""The library fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt) beside ytvln_gemm_f32 on the cfg-2 shapes: a yardstick, not a dependency."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(16128, 1024, 1024, 0, 1), (16128, 3072, 1024, 0, 1), (16128, 1024, 1024, 0, 0), (4480, 768, 3072, 0, 1), (4480, 3072, 768, 0, 1),
          (4480, 2304, 768, 0, 1), (4480, 768, 768, 0, 1), (4480, 768, 3072, 0, 0), (1024, 1024, 16128, 1, 0), (3072, 1024, 16128, 1, 0),
          (768, 3072, 4480, 1, 0), (4480, 30528, 768, 0, 1)]


def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    lda = M if ta else K; ldb = K if tb else N
    mine = timeit(lambda: ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K))
    Ao, Bo = (A.t() if ta else A), (B.t() if tb else B)
    lib = timeit(lambda: torch.matmul(Ao, Bo, out=C))
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:6d} {K:6d} tA{ta} tB{tb}   ytvln {mine:8.1f} us {fl / mine / 1e6:6.1f} TF/s   torch {lib:8.1f} us {fl / lib / 1e6:6.1f} TF/s   ratio {lib / mine:5.2f}", flush=True)
